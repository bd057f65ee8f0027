#!/usr/bin/env python3
"""find_period on the bench workload with environment settings A/B'd in one process (interleaved rounds).

    python scripts/ab_find_period.py PARRM_FIT_PERIODS_BY_COPY=1
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import FS, F_ART, assumed_periods_1e4
from pyparrm_amd import PARRM
from pyparrm_amd.synth import synth_recording_device

arms = [("default", {})] + [(a, dict([a.split("=", 1)])) for a in sys.argv[1:]]
x = synth_recording_device(256, 10_000_000, FS, F_ART, seed=0)
torch.cuda.synchronize()
assumed = assumed_periods_1e4()
times = {name: [] for name, _ in arms}
period = {}
for rnd in range(7):
    for name, env in arms:
        for k, v in env.items():
            os.environ[k] = v
        p = PARRM(x, FS, F_ART, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.find_period(assumed_periods=assumed, random_seed=44)
        torch.cuda.synchronize()
        if rnd:
            times[name].append((time.perf_counter() - t0) * 1e3)
        period[name] = p.period
        for k in env:
            os.environ.pop(k, None)
for name, _ in arms:
    t = sorted(times[name])
    print(f"{name:40s} median {t[len(t) // 2]:7.2f} ms  min {t[0]:7.2f}  max {t[-1]:7.2f}  period {period[name]!r}")
