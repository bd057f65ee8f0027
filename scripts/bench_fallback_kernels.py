#!/usr/bin/env python3
"""Timings of the filter kernels a plan falls back to (stride, gather) beside the phase kernel, and of
half-widths beyond any LDS ring: 64 ch x 1 M float64 (BASELINE configs[1] shape), HIP events, best of 5."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, _hip

_hip.require_gpu()
C, N = 64, 1_000_000
x = torch.randn((C, N), dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
names = {_hip.KERNEL_AUTO: "auto", _hip.KERNEL_GATHER: "gather", _hip.KERNEL_STRIDE: "stride", _hip.KERNEL_PHASE: "phase", 4: "segmented"}
print(f"{'half-width':>10} {'taps':>6} {'auto':>12} | kernel: ms (GB/s of 16 B/sample)")
for hw in (2372, 5000, 9000, 20000, 60000):
    p = PARRM(np.zeros((1, 200_000)), 22000, 130, verbose=False)
    p._period = np.float64(169.23584580707903)
    p.create_filter(filter_half_width=hw)
    plan = _hip.FilterPlan(p.filter)
    info = plan.info
    kernels = [_hip.KERNEL_GATHER, _hip.KERNEL_AUTO]
    if info.stride > 0:
        kernels.append(_hip.KERNEL_STRIDE)
    if info.phase_groups > 0:
        kernels.append(_hip.KERNEL_PHASE)
    out = torch.empty_like(x)
    cells = []
    for k in kernels:
        plan.set_kernel(k)
        plan.apply(x, out=out)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.apply(x, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        cells.append(f"{names[k]}: {best:.3f} ({16 * C * N / best / 1e6:.0f})")
    plan.set_kernel(_hip.KERNEL_AUTO)
    auto = names.get(int(info.kernel), str(int(info.kernel))) + (f"x{int(info.reserved)}" if int(info.reserved) else "")
    print(f"{hw:>10} {int(info.n_taps):>6} {auto:>12} | " + "; ".join(cells), flush=True)
