#!/usr/bin/env python3
"""Run the batched fit objective in isolation (for rocprofv3): stage-3-like shape by default."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--chans", type=int, default=256)
ap.add_argument("--n", type=int, default=24963)
ap.add_argument("--periods", type=int, default=381)
ap.add_argument("--bw", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch
from pyparrm_amd import _hip
_hip.require_gpu()
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn((a.n, a.chans), dtype=torch.float64, device="cuda", generator=g).clamp_(-3, 3)
idx = torch.arange(a.n, dtype=torch.int64, device="cuda") + 1000
per = 169.2358 * (1 + np.linspace(-3e-3, 3e-3, a.periods))
ws = _hip.FitWorkspace()
_hip.fit_errors(y, idx, per, a.bw, 1.0, ws)
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter(); _hip.fit_errors(y, idx, per, a.bw, 1.0, ws); ts.append(time.perf_counter() - t0)
K = 2 * a.bw + 1
flops = 2.0 * a.n * K * (a.chans + K) * a.periods
print(f"P={a.periods} n={a.n} C={a.chans} bw={a.bw}: best {min(ts)*1e3:.3f} ms -> {flops/min(ts)/1e12:.1f} TFLOP/s algorithmic")
