#!/usr/bin/env python3
"""parrm_filter_host on unlocked host arrays below the lock-in-place size (the staged path): chunk size A/B."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyparrm_amd import PARRM, _hip

_hip.require_gpu()
p = PARRM(np.zeros((1, 200_000)), 22000, 130, verbose=False)
p._period = np.float64(169.23584580707903)
p.create_filter()
plan = _hip.FilterPlan(p.filter)
rng = np.random.default_rng(0)
for shape in ((64, 100_000), (8, 1_000_000), (2, 200_000)):
    x = rng.standard_normal(shape)
    out = np.empty_like(x)
    for mb in ("", "256", "4", "64"):
        if mb:
            os.environ["PARRM_HOST_CHUNK_MB"] = mb
        else:
            os.environ.pop("PARRM_HOST_CHUNK_MB", None)
        plan.apply_host(x, out=out)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            plan.apply_host(x, out=out)
            best = min(best, time.perf_counter() - t0)
        print(f"{shape[0]:3d} ch x {shape[1]:8d} f64 ({x.nbytes / 2**20:6.1f} MiB) chunk {mb or 'default':>7} MiB: {best * 1e3:8.2f} ms "
              f"({2 * x.nbytes / best / 1e9:5.1f} GB/s in + out)", flush=True)
os.environ.pop("PARRM_HOST_CHUNK_MB", None)
