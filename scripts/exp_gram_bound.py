#!/usr/bin/env python3
"""What bounds fit_accum_mfma_kernel?  Times the three grids of the bench workload with (a) the W'W tiles left out
(PARRM_FIT_X_NOGRAM: 12 instead of 15 MFMAs per wave and step -- wrong errors, timing only) and (b) forced sample
splits (PARRM_FIT_X_NSPLIT)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyparrm_amd import _hip
_hip.require_gpu()
C = 256
g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(10044, 5001, 5), (381, 24963, 20), (9, 24963, 20), (4, 24963, 20), (12, 5001, 5), (10, 10001, 10)]
arms = [{}] + [{"PARRM_FIT_X_NSPLIT": str(k)} for k in (1, 2, 3, 32, 48, 64)]
for P, n, bw in shapes:
    y = torch.randn((n, C), generator=g, device="cuda", dtype=torch.float64).clamp_(-3, 3)
    idx = torch.arange(1000, 1000 + n, device="cuda", dtype=torch.int64)
    periods = 169.2 * (1 + np.linspace(-1e-2, 1e-2, P))
    line = []
    for arm in arms:
        for k, v in arm.items():
            os.environ[k] = v
        ws = _hip.FitWorkspace()
        for _ in range(2):
            _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
        reps = 5 if P > 100 else 50
        best = 1e9
        for rnd in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        for k in arm:
            os.environ.pop(k)
        line.append(f"{'default' if not arm else ','.join(f'{k[12:]}={v}' for k, v in arm.items())}: {best:.3f}")
        del ws
    print(f"P={P} n={n} bw={bw}: " + "  ".join(line), flush=True)
