#!/usr/bin/env python3
"""Host profile of find_period_batched on eight single-channel sites (the per-site use of examples/plot_example_dbs_data.py)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, find_period_batched
from pyparrm_amd.synth import synth_recording_exact

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sites = [synth_recording_exact(1, 60000, 1000.0 / 130.0 * (1 + 1e-4 * k), seed=40 + k) for k in range(n_sites)]


def batched():
    ps = [PARRM(x, 1000, 130, verbose=False) for x in sites]
    find_period_batched(ps, random_seed=3)
    return [p.period for p in ps]


def sequential():
    out = []
    for x in sites:
        p = PARRM(x, 1000, 130, verbose=False)
        p.find_period(random_seed=3)
        out.append(p.period)
    return out


batched(), sequential()
for name, fn in (("sequential", sequential), ("batched", batched)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {n_sites} sites", flush=True)
pr = cProfile.Profile()
pr.enable()
batched()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
