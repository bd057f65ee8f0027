#!/usr/bin/env python3
"""Wide-toothed filters (period_half_width = period / 8 ... / 12) on 256 ch x 10 M float64: the generated kernel
against the kernel the plan chooses without it (PARRM_COMB=0: the stride kernel -- the phase-major kernel's guard
columns end at 6 residues)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import parrm_oracle as orc
from pyparrm_amd import _hip

_hip.require_gpu()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((256, 10_000_000), dtype=torch.float64, device="cuda", generator=g)
y = torch.empty_like(x)
KNAME = {1: "gather", 2: "stride", 3: "phase", 4: "segmented"}
for period, div in ((169.2359, 50), (169.2359, 12), (169.2359, 8), (123.08, 8), (101.77, 8), (175.9, 8)):
    hw = orc.default_filter_half_width(x.shape[1], period, 0, period / div)
    filt = orc.generate_filter(period, hw, 0, "both", period / div)
    row = [f"T {period:8.4f} phw T/{div:<2d} hw {hw:5d} taps {int((filt != 0).sum()) - 1:5d}"]
    for comb in ("1", "0"):
        os.environ["PARRM_COMB"] = comb
        plan = _hip.FilterPlan(filt)
        plan.apply(x, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            plan.apply(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        st, stride, msg = plan.generated
        name = f"generated (stride {stride})" if st == 1 else KNAME[int(plan.info.kernel)]
        row.append(f"{name:24s} {ms:7.2f} ms = {16 * x.numel() / ms / 1e9 / 8:5.3f} of 8 TB/s")
    print(" | ".join(row), flush=True)
