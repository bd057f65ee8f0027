#!/usr/bin/env python3
"""Randomised end-to-end parity of find_period against the oracle (the reference's algorithm on the
CPU): small random recordings, seeds, search windows, assumed periods, outlier boundaries.

    python scripts/fuzz_period.py --cases 6 --seed 0      # ~10-30 s of CPU per case
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=6)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from oracle import parrm_oracle as orc
    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.synth import synth_recording

    _hip.require_gpu()
    rng = np.random.default_rng(args.seed)
    for case in range(args.cases):
        fs, f_art = (22000.0, 130.0) if rng.random() < 0.6 else (1000.0, 129.3)
        n_chans = int(rng.integers(1, 6))
        n_samples = int(rng.choice([6000, 20000, 40000]))
        x = synth_recording(n_chans, n_samples, fs, f_art, seed=int(rng.integers(1 << 30)))
        kw = {"random_seed": int(rng.integers(0, 1000)), "outlier_boundary": float(rng.choice([3.0, 2.0, 5.5]))}
        if rng.random() < 0.4:
            lo = int(rng.integers(0, n_samples // 4))
            hi = int(rng.integers(3 * n_samples // 4, n_samples))
            kw["search_samples"] = np.arange(lo, hi)
        if rng.random() < 0.4:
            kw["assumed_periods"] = float(fs / f_art * (1 + rng.uniform(-3e-3, 3e-3)))
        t0 = time.perf_counter()
        ref = orc.find_period(x, fs, f_art, **kw)
        t1 = time.perf_counter()
        p = PARRM(x, fs, f_art, verbose=False)
        p.find_period(**kw)
        t2 = time.perf_counter()
        rel = abs(float(p.period) - float(ref)) / abs(float(ref))
        print(f"case {case}: C {n_chans} N {n_samples} fs {fs} {sorted(kw)}: ref {ref!r} gpu {p.period!r} rel {rel:.2e} "
              f"(oracle {t1 - t0:.1f} s, gpu {1e3 * (t2 - t1):.0f} ms)", flush=True)
        if not rel <= 1e-9:
            sys.exit(1)
    print("all periods within 1e-9")


if __name__ == "__main__":
    main()
