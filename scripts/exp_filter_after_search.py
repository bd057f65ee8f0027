#!/usr/bin/env python3
"""Why is the filter kernel 4-7 % slower inside the bench step than back to back?  Times the kernel (library events)
after different predecessors: itself, find_period, find_period + an idle gap, find_period + a streaming read pass."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, _hip
from pyparrm_amd.synth import synth_recording_device

FS, FA = 22000.0, 130.0
assumed = tuple(FS / FA * (1 + 0.02 * k) for k in range(-13, 13))
x = synth_recording_device(256, 10_000_000, FS, FA, seed=0)
p = PARRM(x, FS, FA, verbose=False)
p.find_period(assumed_periods=assumed, random_seed=44)
p.create_filter()
plan = _hip.FilterPlan(p.filter)
y = torch.empty_like(x)


def filt():
    _hip.filter_kernel_timing(True)
    plan.apply(x, out=y)
    torch.cuda.synchronize()
    return _hip.filter_kernel_timing(True)


def search():
    q = PARRM(x, FS, FA, verbose=False)
    q.find_period(assumed_periods=assumed, random_seed=44)


for _ in range(3):
    filt()
res = {k: [] for k in ("back_to_back", "after_search", "second_after_search", "after_search_idle_2ms", "after_search_idle_20ms",
                        "after_search_then_absdiff", "after_absdiff_only", "after_idle_20ms", "after_search_then_touch_8ch",
                        "after_search_then_touch_32ch", "after_search_then_touch_64ch")}
touch_total = {8: [], 32: [], 64: []}
for rep in range(8):
    filt()
    res["back_to_back"].append(filt())
    search()
    res["after_search"].append(filt())
    res["second_after_search"].append(filt())
    search()
    torch.cuda.synchronize()
    time.sleep(0.002)
    res["after_search_idle_2ms"].append(filt())
    search()
    torch.cuda.synchronize()
    time.sleep(0.02)
    res["after_search_idle_20ms"].append(filt())
    search()
    _hip.absdiff_mean(x)
    res["after_search_then_absdiff"].append(filt())
    for rows in (8, 32, 64):  # a short streaming read of the first rows right in front of the launch
        search()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.absdiff_mean(x[:rows])
        res[f"after_search_then_touch_{rows}ch"].append(filt())
        e1.record()
        torch.cuda.synchronize()
        touch_total[rows].append(e0.elapsed_time(e1))
    filt()
    _hip.absdiff_mean(x)
    res["after_absdiff_only"].append(filt())
    filt()
    torch.cuda.synchronize()
    time.sleep(0.02)
    res["after_idle_20ms"].append(filt())
for k, v in res.items():
    v = np.array(v)
    print(f"{k:28s} median {np.median(v):6.3f} ms  min {v.min():6.3f}  max {v.max():6.3f}")
for rows, v in touch_total.items():
    print(f"touch {rows} rows + filter launch, events around both: median {np.median(v):6.3f} ms")
