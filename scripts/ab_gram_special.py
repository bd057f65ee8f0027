#!/usr/bin/env python3
"""W'W from three of its rows (the Gram kernel's GM = 2 form) against the full matrix
(PARRM_FIT_FULL_GRAM=1): largest relative difference of the errors and the time per call, bench shapes and the
narrower kernel forms."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyparrm_amd import _hip
_hip.require_gpu()
g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(10044, 5001, 5), (387, 10001, 10), (381, 24963, 20), (9, 24963, 20), (4, 24963, 20), (12, 5001, 5), (10, 10001, 10)]
for C in (256, 100, 64, 20, 8, 1, 300):
    for P, n, bw in shapes:
        if C != 256 and P > 1000:
            P = 1203
        y = torch.randn((n, C), generator=g, device="cuda", dtype=torch.float64).clamp_(-3, 3)
        idx = torch.sort(torch.randperm(3 * n, generator=g, device="cuda")[:n]).values.to(torch.int64) + 1000
        periods = 169.2 * (1 + np.linspace(-1e-2, 1e-2, P))
        res = {}
        for name in ("special", "full"):
            if name == "full":
                os.environ["PARRM_FIT_FULL_GRAM"] = "1"
            ws = _hip.FitWorkspace()
            for _ in range(2):
                e = _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
            reps = 5 if P > 100 else 50
            best = 1e9
            for rnd in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    _hip.fit_errors(y, idx, periods, bw, 1.0, ws)
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            os.environ.pop("PARRM_FIT_FULL_GRAM", None)
            res[name] = (np.asarray(e), best)
        d = np.abs(res["special"][0] - res["full"][0]).max() / np.abs(res["full"][0]).max()
        print(f"C={C:3d} P={P:5d} n={n:5d} bw={bw:2d}: full {res['full'][1]:7.3f} ms  special {res['special'][1]:7.3f} ms   "
              f"max rel diff {d:.2e}  finite {int(np.isfinite(res['special'][0]).sum())}/{P}", flush=True)
