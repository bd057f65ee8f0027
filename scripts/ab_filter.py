#!/usr/bin/env python3
"""A/B the phase-kernel shapes (PARRM_PHASE_SHAPE) in ONE process on one device, interleaved, so
that box-to-box and clock drift do not masquerade as a kernel difference.

    python scripts/ab_filter.py --shapes 4,2 2,4 2,3 --rounds 7
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
    ap.add_argument("--shapes", nargs="+", default=["4,2", "2,4"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--flags", nargs="*", default=None, help="PARRM_DEBUG_FLAGS values to A/B instead of shapes")
    ap.add_argument("--stretch", nargs="*", default=None, help="PARRM_STRETCH_SAMPLES values to A/B (0 = the plan's own choice)")
    ap.add_argument("--plan-env", nargs="*", default=None, help="NAME=VALUE settings in force while the plan is BUILT, one arm each")
    ap.add_argument("--env", nargs="*", default=None, help="NAME=VALUE settings to A/B against the default")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--lib", default=None, help="load this build of libparrm_hip.so instead of the in-tree one")
    args = ap.parse_args()

    import torch

    from pyparrm_amd import PARRM, _hip

    if args.lib:
        _hip._LIB_PATH = os.path.abspath(args.lib)
    _hip.require_gpu()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((args.chans, args.samples), dtype=torch.float64, device="cuda", generator=g)
    if args.dtype == "f32":
        x = x.to(torch.float32)
    y = torch.empty_like(x)
    p = PARRM(np.zeros((1, args.samples)), 22000.0, 130.0, verbose=False)
    p._period = np.float64(22000.0 / 130.0 * (1 + 3e-5))
    p.create_filter()
    arms = []
    if args.plan_env:
        for kv in args.plan_env:
            k, v = kv.split("=", 1)
            os.environ[k] = v
            plan = _hip.FilterPlan(p.filter)
            os.environ.pop(k, None)
            info = plan.info
            arms.append((f"plan {kv} (residues/lane {info.phase_residues}, NG={info.phase_groups}, M={info.phase_row_slots})", plan, {}))
    elif args.env:
        plan = _hip.FilterPlan(p.filter)
        arms.append(("default", plan, {}))
        for kv in args.env:
            k, v = kv.split("=", 1)
            arms.append((kv, plan, {k: v}))
    elif args.stretch:
        plan = _hip.FilterPlan(p.filter)
        for v in args.stretch:
            arms.append((f"stretch={v}", plan, {} if v == "0" else {"PARRM_STRETCH_SAMPLES": v}))
    elif args.flags:
        plan = _hip.FilterPlan(p.filter)
        plan.set_kernel(3)
        for f in args.flags:
            arms.append((f"flags={f}", plan, {"PARRM_DEBUG_FLAGS": f}))
    else:
        for sh in args.shapes:
            os.environ["PARRM_PHASE_SHAPE"] = sh
            plan = _hip.FilterPlan(p.filter)
            plan.set_kernel(3)
            info = plan.info
            arms.append((f"shape={sh} (NG={info.phase_groups} R={info.phase_rows} M={info.phase_row_slots})", plan, {}))
        os.environ.pop("PARRM_PHASE_SHAPE", None)
    times = {name: [] for name, _, _ in arms}
    for rnd in range(args.rounds + 1):
        for name, plan, env in arms:
            for k, v in env.items():
                os.environ[k] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.apply(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            for k in env:
                os.environ.pop(k, None)
            if rnd:  # round 0 warms up
                times[name].append(e0.elapsed_time(e1))
    nbytes = 2 * x.numel() * x.element_size()
    for name, ts in times.items():
        med = float(np.median(ts))
        print(f"{name}: median {med:.3f} ms  min {min(ts):.3f}  max {max(ts):.3f}  -> {nbytes / med / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
