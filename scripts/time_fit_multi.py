#!/usr/bin/env python3
"""Wall time of one parrm_fit_errors_multi call (one optimiser step of several independent searches)."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import _hip

n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(1)
for P, n, bw in ((9, 24963, 20), (3, 24963, 20), (12, 5001, 5), (10, 10001, 10)):
    items = []
    for k in range(n_prob):
        y = torch.randn((n, (C + 3) // 4 * 4), generator=g, device="cuda", dtype=torch.float64).clamp_(-3, 3)[:, :C]
        idx = torch.arange(1000, 1000 + n, device="cuda", dtype=torch.int64)
        periods = 7.69 * (1 + np.linspace(-1e-2, 1e-2, P) + 1e-4 * k)
        items.append((SimpleNamespace(y=y, d_idx=idx, periods=periods, bandwidth=bw, lambda_=1.0), _hip.FitWorkspace()))
    line = f"{n_prob} problems x (P={P}, n={n}, bw={bw}, C={C}):"
    for mode in ("gang", "streams"):
        if mode == "streams":
            os.environ["PARRM_FIT_MULTI_STREAMS"] = "1"
        else:
            os.environ.pop("PARRM_FIT_MULTI_STREAMS", None)
        for _ in range(5):
            out = _hip.fit_errors_multi(items)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            out = _hip.fit_errors_multi(items)
        dt = (time.perf_counter() - t0) / 30
        line += f"  {mode} {dt * 1e3:.3f} ms"
    single = time.perf_counter()
    for _ in range(10):
        for req, ws in items:
            _hip.fit_errors(req.y, req.d_idx, req.periods, req.bandwidth, req.lambda_, ws)
    line += f"  one-by-one {(time.perf_counter() - single) / 10 * 1e3:.3f} ms"
    print(line, flush=True)
os.environ.pop("PARRM_FIT_MULTI_STREAMS", None)
