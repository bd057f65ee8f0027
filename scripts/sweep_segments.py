#!/usr/bin/env python3
"""Window half-width of segmented plans (PARRM_SEGMENT_HALFWIDTH) against launch time, 64 ch x 1 M float64."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, _hip

_hip.require_gpu()
C, N = 64, 1_000_000
x = torch.randn((C, N), dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
out = torch.empty_like(x)
for hw in (9000, 20000, 60000):
    p = PARRM(np.zeros((1, 200_000)), 22000, 130, verbose=False)
    p._period = np.float64(169.23584580707903)
    p.create_filter(filter_half_width=hw)
    for H in ("", "16000", "12000", "9600", "8000", "6400", "4800", "2400"):
        if H:
            os.environ["PARRM_SEGMENT_HALFWIDTH"] = H
        else:
            os.environ.pop("PARRM_SEGMENT_HALFWIDTH", None)
        plan = _hip.FilterPlan(p.filter)
        info = plan.info
        if int(info.kernel) != 4:
            print(f"hw {hw} H {H or 'auto'}: not segmented", flush=True)
            continue
        plan.apply(x, out=out)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.apply(x, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"hw {hw} H {H or 'auto':>5}: {int(info.reserved):3d} passes {best:8.3f} ms  ({best / int(info.reserved):.3f} per pass)", flush=True)
os.environ.pop("PARRM_SEGMENT_HALFWIDTH", None)
