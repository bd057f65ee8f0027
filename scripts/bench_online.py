#!/usr/bin/env python3
"""Per-block cost of OnlineFilter (block-by-block filter_data): 64 ch, 22 kHz, default filter geometry."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, _hip
from pyparrm_amd.streaming import OnlineFilter

_hip.require_gpu()
C = 64
for direction in ("future", "both"):
    p = PARRM(np.zeros((1, 200_000)), 22000, 130, verbose=False)
    p._period = np.float64(169.23584580707903)
    p.create_filter(filter_direction=direction)
    for block in (2200, 22000, 220000):  # 0.1 s, 1 s, 10 s of signal
        stream = OnlineFilter(p.filter, C)
        x = torch.randn((C, block), dtype=torch.float64, device="cuda")
        for _ in range(5):
            stream.push(x)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            y = stream.push(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{direction:6s} latency {stream.latency:5d} samples | block {block:6d} samples x {C} ch: {dt * 1e3:7.3f} ms per push "
              f"= {C * block / dt / 1e6:8.1f} Msamples/s  (real time at 22 kHz needs {C * 22000 / 1e6:.2f})", flush=True)
