#!/usr/bin/env python3
"""Where do the ~27 us between two optimiser batches go?  Times the C call of every small batch and the Python time
between two calls (Nelder-Mead bookkeeping, request plumbing)."""
import os, sys, time
os.environ.setdefault("PARRM_NM_PYTHON", "1")  # (the refinement stepped from Python: inside the library there is no call to time)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyparrm_amd import PARRM, _hip
from pyparrm_amd.synth import synth_recording_device
_hip.require_gpu()
x = synth_recording_device(256, 10_000_000, 22000.0, 130.0, seed=0)
base = 22000.0 / 130.0
assumed = tuple(base * (1 + 0.02 * k) for k in range(-13, 13))
orig = _hip.FitWorkspace.small_batch
calls = []
def timed(self, *a, **k):
    t0 = time.perf_counter(); out = orig(self, *a, **k); t1 = time.perf_counter()
    calls.append((t0, t1)); return out
_hip.FitWorkspace.small_batch = timed
for rep in range(4):
    calls.clear()
    p = PARRM(x, 22000.0, 130.0, verbose=False)
    t0 = time.perf_counter(); p.find_period(assumed_periods=assumed, random_seed=44); t1 = time.perf_counter()
    inside = sum(b - a for a, b in calls)
    between = [calls[i + 1][0] - calls[i][1] for i in range(len(calls) - 1)]
    small = [g for g in between if g < 200e-6]
    print(f"find_period {1e3 * (t1 - t0):.2f} ms; {len(calls)} small batches: inside the C call {1e3 * inside:.2f} ms, "
          f"python between consecutive batches: median {1e6 * np.median(small):.1f} us, sum {1e3 * sum(small):.2f} ms "
          f"({len(between) - len(small)} longer gaps = stage boundaries: {1e3 * (sum(between) - sum(small)):.2f} ms)", flush=True)
