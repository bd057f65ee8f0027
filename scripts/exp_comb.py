#!/usr/bin/env python3
"""First-light / A-B script for the generated comb kernel: parity against the gather kernel (tap-by-tap
evaluation) on awkward shapes, then interleaved timing against the phase-major kernel on the headline shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PARRM_COMB_VERBOSE", "1")


def main():
    import torch

    from pyparrm_amd import PARRM, _hip

    _hip.require_gpu()
    p = PARRM(np.zeros((1, 10_000_000)), 22000.0, 130.0, verbose=False)
    p._period = np.float64(22000.0 / 130.0 * (1 + 3e-5))
    p.create_filter()
    filt = p.filter
    t0 = time.time()
    plan = _hip.FilterPlan(filt)
    gather = _hip.FilterPlan(filt)
    gather.set_kernel(1)
    g = torch.Generator(device="cuda").manual_seed(1)
    ok = True
    for (c, n) in [(3, 40_000), (2, 300_001), (5, 1_000_003), (1, 2_500_000)]:
        x = torch.randn((c, n), dtype=torch.float64, device="cuda", generator=g)
        os.environ["PARRM_COMB"] = "force"
        y = plan.apply(x)
        os.environ["PARRM_COMB"] = "0"
        y_old = plan.apply(x)
        y_ref = gather.apply(x)
        torch.cuda.synchronize()
        e_new = (y - y_ref).abs().max().item()
        e_old = (y_old - y_ref).abs().max().item()
        bad = int((~torch.isfinite(y)).sum().item())
        print(f"shape {c} x {n}: comb vs gather {e_new:.3e}   phase vs gather {e_old:.3e}   non-finite {bad}", flush=True)
        if not (e_new < 1e-9) or bad:
            ok = False
            d = (y - y_ref).abs()
            idx = torch.nonzero(d > 1e-9)
            print("   first bad:", idx[:8].tolist(), " count", idx.shape[0], flush=True)
    # window form: outputs [o0, o0+len) from a buffer with halo
    x = torch.randn((2, 700_000), dtype=torch.float64, device="cuda", generator=g)
    y_ref = gather.apply(x)
    os.environ["PARRM_COMB"] = "force"
    hw = plan.info.half_width
    o0, ol = 123_457, 400_001
    b0 = o0 - hw - 5
    buf = x[:, b0:o0 + ol + hw + 3].contiguous()
    yw = plan.apply_window(buf, b0, o0, ol, x.shape[1])
    torch.cuda.synchronize()
    e = (yw - y_ref[:, o0:o0 + ol]).abs().max().item()
    print(f"window form: {e:.3e}", flush=True)
    ok = ok and e < 1e-9
    print("first-light parity:", "OK" if ok else "FAILED", f"({time.time() - t0:.1f} s)", flush=True)
    if not ok and "--force-timing" not in sys.argv:
        return 1
    # timing, interleaved
    c, n = 256, 10_000_000
    x = torch.randn((c, n), dtype=torch.float64, device="cuda", generator=g)
    y = torch.empty_like(x)
    arms = [("phase", "0"), ("comb", "force")]
    times = {k: [] for k, _ in arms}
    for rnd in range(6):
        for name, env in arms:
            os.environ["PARRM_COMB"] = env
            plan.apply(x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                plan.apply(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 3)
    for name, _ in arms:
        t = np.array(times[name][1:])
        print(f"{name:6s} {t.min():7.3f} ms min  {np.median(t):7.3f} median   -> {16 * c * n / t.min() / 1e9:6.3f} TB/s  frac {16 * c * n / t.min() / 1e9 / 8:5.3f}", flush=True)
    os.environ["PARRM_COMB"] = "force"
    yc = plan.apply(x)
    os.environ["PARRM_COMB"] = "0"
    yp = plan.apply(x)
    print("headline comb vs phase max |d|:", (yc - yp).abs().max().item(), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
