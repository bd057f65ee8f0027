#!/usr/bin/env python3
"""Randomised parity sweep of filter_data against the closed form (oracle), across recording shapes,
periods, half-widths, directions, dtypes and workgroup shapes -- the edges of the interior fast
path (stretch seams, recording ends, tiny recordings) are where a bug would hide.

    python scripts/fuzz_filter.py --cases 120 --seed 0
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch

    from oracle import parrm_oracle as orc
    from pyparrm_amd import _hip

    _hip.require_gpu()
    rng = np.random.default_rng(args.seed)
    worst = 0.0
    for case in range(args.cases):
        period = float(rng.choice([7.7424, 13.0, 64.3, 101.77, 169.2359, 169.5, 250.01, 333.3, 480.9]))
        period *= 1 + rng.uniform(-2e-3, 2e-3)
        n_chans = int(rng.integers(1, 7))
        n_samples = int(rng.choice([1, 2, 17, 500, 5000, 40_000, 131_073, 300_000, 700_001]))
        hw_max = max(1, (n_samples - 1) // 2)
        hw = int(min(hw_max, rng.choice([3, 50, 650, 2372, 6000])))
        omit = int(rng.integers(0, max(1, min(hw, 30))))
        direction = str(rng.choice(["both", "past", "future"]))
        phw = float(period / rng.choice([50, 20, 8]))
        try:
            taps = orc.generate_filter(period, hw, omit, direction, phw)
        except RuntimeError:
            continue
        dtype = np.float64 if rng.random() < 0.7 else np.float32
        x = rng.standard_normal((n_chans, n_samples)).astype(dtype)
        shape = str(rng.choice(["", "4,2", "2,4", "2,3", "3,2", "2,2", "1,4"]))
        stretch = str(rng.choice(["", "20000", "60000"]))
        for key, val in (("PARRM_PHASE_SHAPE", shape), ("PARRM_STRETCH_SAMPLES", stretch)):
            if val:
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        plan = _hip.FilterPlan(taps)
        ref = orc.filter_data_direct(x.astype(np.float64), taps)
        scale = max(float(np.abs(ref).max()), 1e-300)
        tol = 1e-10 if dtype == np.float64 else 1e-5
        kernels = [_hip.KERNEL_AUTO, _hip.KERNEL_GATHER]
        if plan.info.stride > 0:
            kernels.append(_hip.KERNEL_STRIDE)
        if plan.info.phase_groups > 0:
            kernels.append(_hip.KERNEL_PHASE)
        for kern in kernels:
            plan.set_kernel(kern)
            y = plan.apply(torch.from_numpy(x).cuda()).cpu().numpy()
            err = float(np.abs(y - ref).max()) / scale if y.size else 0.0
            worst = max(worst, err if dtype == np.float64 else 0.0)
            if not err <= tol:
                print(f"FAIL case {case}: period {period} C {n_chans} N {n_samples} hw {hw} omit {omit} {direction} "
                      f"phw {phw} {dtype.__name__} shape '{shape}' stretch '{stretch}' kernel {kern}: rel err {err:.3e}")
                sys.exit(1)
        plan.set_kernel(_hip.KERNEL_AUTO)
        if case % 10 == 9:
            print(f"  ... {case + 1} cases, worst so far {worst:.2e}", flush=True)
    print(f"{args.cases} cases ok; worst f64 relative error {worst:.2e}")


if __name__ == "__main__":
    main()
