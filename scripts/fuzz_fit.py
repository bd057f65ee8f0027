#!/usr/bin/env python3
"""Randomised parity sweep of the harmonic-regression errors (parrm_fit_errors) against the oracle:
sample counts, channel counts (column-quad padding), bandwidths (fast and generic solvers), lambda,
batch sizes (sample splits, host hand-off vs copy path).

    python scripts/fuzz_fit.py --cases 80 --seed 0
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch

    from oracle import parrm_oracle as orc
    from pyparrm_amd import _hip
    from pyparrm_amd.synth import synth_recording

    _hip.require_gpu()
    rng = np.random.default_rng(args.seed)
    worst = 0.0
    for case in range(args.cases):
        n_chans = int(rng.integers(1, 10)) if rng.random() < 0.8 else int(rng.choice([64, 257, 300]))
        n_samples = int(rng.choice([4000, 9000, 30000]))
        x = synth_recording(n_chans, n_samples, 22000, 130, seed=int(rng.integers(1 << 30)))
        std = orc.standardise_data(x, 3.0)
        # (at least ~4 periods of samples: with fewer the harmonics are collinear, W'W is numerically
        # singular and neither LAPACK nor this solver returns anything meaningful to compare)
        n_idx = int(rng.choice([700, 1000, 2500, 3999])) if n_samples == 4000 else int(rng.integers(700, n_samples - 2))
        n_idx = min(n_idx, n_samples - 3)
        if rng.random() < 0.5:
            lo = int(rng.integers(0, n_samples - 1 - n_idx))
            idx = np.arange(lo, lo + n_idx)
        else:
            idx = np.unique(rng.integers(0, n_samples - 1, n_idx))
        bw = int(rng.choice([1, 3, 5, 7, 10, 15, 20, 23]))
        bw = min(bw, idx.shape[0] // 4)  # the reference's clipping (parrm.py:437, :497)
        if bw < 1:
            continue
        lam = float(rng.choice([0.0, 0.5, 1.0]))
        n_per = int(rng.choice([1, 2, 5, 19, 60, 130]))
        # (the oracle solves one least-squares problem per channel and period: bound its cost, so that a wide case does
        # not keep the run silent for minutes -- a 300-channel x 130-period x 30 000-sample case once got a GPU run killed
        # by the pool's silence guard)
        while n_chans * n_per * idx.shape[0] * (2 * bw + 1) > 4e9 and n_per > 1:
            n_per = max(1, n_per // 2)
        periods = 169.2359 * (1 + rng.uniform(-1e-2, 1e-2, n_per))
        d = torch.from_numpy(x).cuda()
        d_idx = torch.from_numpy(idx).cuda()
        y = _hip.gather_standardise(d, d_idx, _hip.absdiff_mean(d), 3.0)
        ref = orc.grid_errors(periods, std, idx, bw, lam)
        for env in ({}, {"PARRM_FIT_ACCUM": "1"}, {"PARRM_FIT_COPY_PATH": "1"}):
            for k, v in env.items():
                os.environ[k] = v
            out = _hip.fit_errors(y, d_idx, periods, bw, lam)
            for k in env:
                os.environ.pop(k, None)
            fin = np.isfinite(ref)
            err = float(np.max(np.abs(out[fin] - ref[fin]) / np.abs(ref[fin]))) if fin.any() else 0.0
            worst = max(worst, err)
            if not (err <= 1e-8 and np.array_equal(np.isfinite(out), fin)):
                print(f"FAIL case {case}: C {n_chans} N {n_samples} n_idx {idx.shape[0]} bw {bw} lambda {lam} P {n_per} env {env}: "
                      f"rel err {err:.3e}")
                sys.exit(1)
        print(f"  case {case}: C {n_chans} n_idx {idx.shape[0]} bw {bw} P {n_per}: worst so far {worst:.2e}", flush=True)
    print(f"{args.cases} cases ok; worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
