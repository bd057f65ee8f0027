#!/usr/bin/env python3
"""Wall time of the find_period objective on the bench shapes, stage by stage, outside the search
(random stage matrix; the errors are not looked at): the three grids and optimiser-sized batches."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import _hip

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(10044, 5001, 5), (387, 10001, 10), (381, 24963, 20), (9, 24963, 20), (4, 24963, 20), (12, 5001, 5), (10, 10001, 10)]
ws = _hip.FitWorkspace()
arms = [a for a in sys.argv[2:]]  # NAME=VALUE settings to time beside the default (small batches only)
for P, n, bw in shapes:
    y = torch.randn((n, C), generator=g, device="cuda", dtype=torch.float64).clamp_(-3, 3)
    idx = torch.arange(1000, 1000 + n, device="cuda", dtype=torch.int64)
    periods = 169.2 * (1 + np.linspace(-1e-2, 1e-2, P))
    for _ in range(3):
        _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
    reps = 5 if P > 100 else 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e = _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
    dt = (time.perf_counter() - t0) / reps * 1e3
    extra = ""
    if P <= 100:
        for arm in arms:
            k, v = arm.split("=", 1)
            best = {}
            for rnd in range(6):  # interleaved rounds
                for name in ("default", arm):
                    if name != "default":
                        os.environ[k] = v
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
                    best[name] = min(best.get(name, 1e9), (time.perf_counter() - t0) / reps * 1e3)
                    os.environ.pop(k, None)
            extra += f"   [default {best['default']:.3f} vs {arm} {best[arm]:.3f}]"
    print(f"P={P:6d} n={n:6d} bw={bw:2d} C={C}: {dt:8.3f} ms per call   (finite errors: {int(np.isfinite(e).sum())}){extra}", flush=True)
