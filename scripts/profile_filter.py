#!/usr/bin/env python3
"""Launch filter_data kernels in isolation (for rocprofv3 / A-B timing).

    python scripts/profile_filter.py --chans 256 --samples 10000000 --kernel phase --reps 5
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chans", type=int, default=256)
    ap.add_argument("--samples", type=int, default=10_000_000)
    ap.add_argument("--kernel", default="auto", choices=["auto", "gather", "stride", "phase"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--period", type=float, default=22000.0 / 130.0 * (1 + 3e-5))
    args = ap.parse_args()

    import torch

    from pyparrm_amd import PARRM, _hip

    _hip.require_gpu()
    dt = torch.float64 if args.dtype == "f64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((args.chans, args.samples), dtype=dt, device="cuda", generator=g)
    p = PARRM(np.zeros((1, args.samples)), 22000.0, 130.0, verbose=False)
    p._period = np.float64(args.period)
    p.create_filter()
    plan = _hip.FilterPlan(p.filter)
    info = plan.info
    plan.set_kernel({"auto": 0, "gather": 1, "stride": 2, "phase": 3}[args.kernel])
    y = torch.empty((args.chans, args.samples), dtype=torch.float64 if args.dtype == "f64" else dt, device="cuda")
    plan.apply(x, out=y)
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.apply(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    nbytes = args.chans * args.samples * (x.element_size() + y.element_size())
    best, med = min(times), float(np.median(times))
    print(f"kernel={args.kernel} auto->{info.kernel} hw={info.half_width} taps={info.n_taps} "
          f"stride q={info.stride} nd={info.n_delta} | phase q={info.phase_stride} d={info.phase_delta} "
          f"B={info.phase_guard} NG={info.phase_groups} R={info.phase_rows} M={info.phase_row_slots}")
    print(f"{args.chans}x{args.samples} {args.dtype}: median {med:.3f} ms  best {best:.3f} ms  "
          f"-> {nbytes / med / 1e6:.1f} GB/s algorithmic (median)")


if __name__ == "__main__":
    main()
