#!/usr/bin/env python3
"""Which filter geometries does the generated kernel take, refuse, or FAIL its self-test on?  (The self-test keeps a
wrong kernel from ever being used; a failure is a generator defect to fix.)"""
import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARRM_COMB"] = "force"
import torch
from oracle import parrm_oracle as orc
from pyparrm_amd import _hip
_hip.require_gpu()
x = torch.randn((2, 200_000), dtype=torch.float64, device="cuda")
stats = {}
for period, hw, omit, direction, div in itertools.product([101.77, 123.08, 138.5, 169.2359, 169.5, 175.9], [650, 2372, 6000], [0, 7, 29],
                                                         ["both", "past", "future"], [50, 20, 8]):
    try:
        taps = orc.generate_filter(period, hw, omit, direction, period / div)
    except RuntimeError:
        continue
    plan = _hip.FilterPlan(taps)
    plan.apply(x)
    state, stride, msg = plan.generated
    key = "in use" if state == 1 else ("self-test FAILED" if "self-test" in msg else (msg or "declined by the generator"))
    stats.setdefault(key, []).append((period, hw, omit, direction, div, stride))
for k, v in stats.items():
    print(f"{len(v):4d}  {k}")
    if "FAILED" in k or "scratch" in k:
        for item in v[:12]:
            print("        ", item)
