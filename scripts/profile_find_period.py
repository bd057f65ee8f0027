#!/usr/bin/env python3
"""Host-side profile of find_period on the bench workload (cProfile + wall split)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import FS, F_ART, assumed_periods_1e4
from pyparrm_amd import PARRM, _hip
from pyparrm_amd.synth import synth_recording_device

chans = int(sys.argv[1]) if len(sys.argv) > 1 else 256
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
x = synth_recording_device(chans, samples, FS, F_ART, seed=0)
torch.cuda.synchronize()
for _ in range(2):
    p = PARRM(x, FS, F_ART, verbose=False)
    t0 = time.perf_counter()
    p.find_period(assumed_periods=assumed_periods_1e4(), random_seed=44)
    print("find_period wall ms", (time.perf_counter() - t0) * 1e3, "period", p.period)
calls = []
orig = _hip.fit_errors


def wrapped(y, idx, periods, bw, lam, ws=None, grid_periods=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(y, idx, periods, bw, lam, ws)
    calls.append((len(periods), bw, y.shape[0], (time.perf_counter() - t0) * 1e3))
    return out


_hip.fit_errors = wrapped
# (the per-batch split needs the refinement stepped from Python: inside the library -- the default, timed above -- there
# is no call per batch to wrap; the stepped form costs ~13 us more host time per batch)
from pyparrm_amd import parrm as _facade

_facade._NM_IN_PYTHON = True
print("instrumented pass: refinement stepped from Python (PARRM_NM_PYTHON=1's path)")
import pyparrm_amd.parrm as pm

pm._hip.fit_errors = wrapped
p = PARRM(x, FS, F_ART, verbose=False)
t0 = time.perf_counter()
p.find_period(assumed_periods=assumed_periods_1e4(), random_seed=44)
print("instrumented wall ms", (time.perf_counter() - t0) * 1e3)
tot = sum(c[3] for c in calls)
print("fit_errors calls", len(calls), "total ms", tot)
big = [c for c in calls if c[0] > 100]
print("grid calls:", big)
small = [c for c in calls if c[0] <= 100]
for bw in sorted({c[1] for c in small}):
    sel = [c for c in small if c[1] == bw]
    print(f"bw={bw}: {len(sel)} NM batches, mean P {sum(c[0] for c in sel)/len(sel):.1f}, mean ms {sum(c[3] for c in sel)/len(sel):.3f}, total {sum(c[3] for c in sel):.1f}")
_hip.fit_errors = orig
pm._hip.fit_errors = orig
pr = cProfile.Profile()
p = PARRM(x, FS, F_ART, verbose=False)
pr.enable()
p.find_period(assumed_periods=assumed_periods_1e4(), random_seed=44)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
