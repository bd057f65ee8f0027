#!/usr/bin/env python3
"""Which part of find_period slows the filter launch behind it?  The headline filter kernel timed after: itself; the
statistics pass; the stage-1 grid alone (5 ms of float64 matrix-core work); a grid on a tiny stage matrix; 300 tiny
launches (an optimiser-like chain on 8 channels); the whole search."""
import os, sys, time
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
tr = p._trace
idx1 = torch.from_numpy(tr[0]["indices"]).cuda()
scale = _hip.absdiff_mean(x)
y1 = _hip.gather_standardise(x, idx1, scale, 3.0)
grid1 = tr[0]["grid"]
idx3 = torch.from_numpy(tr[2]["indices"]).cuda()
y3 = _hip.gather_standardise(x, idx3, scale, 3.0)
grid3 = tr[2]["grid"]
y_small = y3[:, :8].contiguous()
ws, ws2 = _hip.FitWorkspace(), _hip.FitWorkspace()


def filt():
    _hip.filter_kernel_timing(True)
    plan.apply(x, out=y)
    torch.cuda.synchronize()
    return _hip.filter_kernel_timing(True)


def search():
    q = PARRM(x, FS, FA, verbose=False)
    q.find_period(assumed_periods=assumed, random_seed=44)


pre = {
    "filter": lambda: None,
    "absdiff": lambda: _hip.absdiff_mean(x),
    "grid1 (10044 cand., K=11, 5.3 ms MFMA)": lambda: _hip.fit_errors(y1, idx1, grid1, 5, 1.0, ws),
    "grid3 (381 cand., K=41, 3.7 ms MFMA)": lambda: _hip.fit_errors(y3, idx3, grid3, 20, 1.0, ws),
    "grid1 x3 (16 ms MFMA)": lambda: [_hip.fit_errors(y1, idx1, grid1, 5, 1.0, ws) for _ in range(3)],
    "80 optimiser batches (9 cand., K=41, 256 ch)": lambda: [_hip.fit_errors(y3, idx3, grid3[:9], 20, 1.0, ws) for _ in range(80)],
    "80 optimiser batches on 8 channels": lambda: [_hip.fit_errors(y_small, idx3, grid3[:9], 20, 1.0, ws2) for _ in range(80)],
    "search": search,
}
side = torch.cuda.Stream()
z = torch.empty_like(x[:64])


def batches_with_traffic(y_mat, w):
    """the 80 batches with a streaming copy of 64 rows (10 GB of traffic, ~2 ms each) looping on a side stream beside them"""
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            z.copy_(x[:64])
    for _ in range(80):
        _hip.fit_errors(y_mat, idx3, grid3[:9], 20, 1.0, w)
    torch.cuda.current_stream().wait_stream(side)


pre["80 batches on 8 channels + streaming copies beside them"] = lambda: batches_with_traffic(y_small, ws2)
pre["80 batches (256 ch) + streaming copies beside them"] = lambda: batches_with_traffic(y3, ws)
res = {k: [] for k in pre}
for _ in range(3):
    filt()
for rep in range(7):
    for k, f in pre.items():
        filt(); filt()
        f()
        res[k].append(filt())
for k, v in res.items():
    v = np.array(v)
    print(f"after {k:48s} median {np.median(v):6.3f} ms  min {v.min():6.3f}  max {v.max():6.3f}")
