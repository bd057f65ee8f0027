#!/usr/bin/env python3
"""Why is the filter launch ~1 ms slower inside a bench step than back to back?  Times the same launch
(HIP events on its stream) in different contexts: after idle, after find_period, after a long
HBM-streaming copy, twice in a row after find_period.

    python scripts/cold_start.py > gpurun_out/cold_start.log
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyparrm_amd import PARRM, _hip  # noqa: E402
from pyparrm_amd.synth import synth_recording_device  # noqa: E402

FS, FA = 22000.0, 130.0
_hip.require_gpu()
x = synth_recording_device(256, 10_000_000, FS, FA, seed=0)
y = torch.empty_like(x)
scratch = torch.empty_like(x)
assumed = tuple(FS / FA * (1 + 0.02 * k) for k in range(-13, 13))
p = PARRM(x, FS, FA, verbose=False)
p.find_period(assumed_periods=assumed, random_seed=44)
p.create_filter()
plan = _hip.FilterPlan(p.filter)


def launch():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.apply(x, out=y)
    e1.record()
    return e0, e1


def ms(ev):
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1])


def report(name, values):
    print(f"{name:<58} {np.median(values):7.3f} ms (min {min(values):.3f}, max {max(values):.3f}, n={len(values)})", flush=True)


for _ in range(3):
    ms(launch())
report("back to back (5 launches queued)", [ms(e) for e in [launch() for _ in range(5)]])
vals = []
for _ in range(5):
    torch.cuda.synchronize()
    time.sleep(0.2)
    vals.append(ms(launch()))
report("after 200 ms idle", vals)
vals = []
for _ in range(5):
    torch.cuda.synchronize()
    time.sleep(0.02)
    vals.append(ms(launch()))
report("after 20 ms idle", vals)
vals, second = [], []
for _ in range(5):
    q = PARRM(x, FS, FA, verbose=False)
    q.find_period(assumed_periods=assumed, random_seed=44)
    a = launch()
    b = launch()
    vals.append(ms(a))
    second.append(ms(b))
report("right after find_period (1st launch)", vals)
report("right after find_period (2nd launch, queued behind)", second)
vals = []
for _ in range(5):
    q = PARRM(x, FS, FA, verbose=False)
    q.find_period(assumed_periods=assumed, random_seed=44)
    scratch.copy_(x)  # ~8 ms of pure HBM streaming in front
    vals.append(ms(launch()))
report("after find_period + a 20 GB device copy", vals)
vals = []
for _ in range(5):
    torch.cuda.synchronize()
    time.sleep(0.2)
    scratch.copy_(x)
    vals.append(ms(launch()))
report("after 200 ms idle + a 20 GB device copy", vals)
vals = []
for _ in range(5):
    q = PARRM(x, FS, FA, verbose=False)
    q._standardise_data()  # only the 3 ms statistics pass
    vals.append(ms(launch()))
report("after the statistics pass only", vals)
