#!/usr/bin/env python3
"""filter_data kernel time across artefact periods (one process, default create_filter settings)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import PARRM, _hip

_hip.require_gpu()
chans, samples = 256, 10_000_000
x = torch.randn((chans, samples), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
periods = [float(t) for t in sys.argv[1:]] or list(np.round(np.exp(np.random.default_rng(0).uniform(np.log(1.2), np.log(400), 30)), 4))
for T in periods:
    p = PARRM(np.zeros((1, samples)), 22000.0, 130.0, verbose=False)
    p._period = np.float64(T)
    try:
        p.create_filter()
    except Exception as exc:  # noqa: BLE001
        print(f"T={T}: create_filter: {exc}")
        continue
    if p.filter.shape[0] > 200_001:
        print(f"T={T}: half-width {p._filter_half_width} (taps too rare), skipped")
        continue
    plan = _hip.FilterPlan(p.filter)
    i = plan.info
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.apply(x, out=y); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"T={T:9.4f} hw={i.half_width:5d} taps={i.n_taps:4d} kernel={i.kernel} q={i.phase_stride:3d} d={i.phase_delta:2d} B={i.phase_guard} "
          f"NG={i.phase_groups} R={i.phase_rows} M={i.phase_row_slots:3d}: {np.median(ts[1:]):6.2f} ms  generated {plan.generated}", flush=True)
