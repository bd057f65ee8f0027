#!/usr/bin/env python3
"""Relative difference between the device's fit errors and the reference's own (golden fixture), in the trig
mode the environment selects (PARRM_FIT_EXACT_TRIG)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import _hip, get_example_data_paths

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fit_errors_ecog.npz"))
ecog = np.load(get_example_data_paths("ecog_lfp_data"))
d = torch.from_numpy(ecog).cuda()
scale = _hip.absdiff_mean(d)


def stage(indices):
    d_idx = torch.from_numpy(np.ascontiguousarray(indices)).cuda()
    return _hip.gather_standardise(d, d_idx, scale, 3.0), d_idx


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


y, d_idx = stage(g["idx1"])
y3, d_idx3 = stage(g["idx3"])
print("mode", os.environ.get("PARRM_FIT_EXACT_TRIG", "0"),
      "| K=11 grid", f"{rel(_hip.fit_errors(y, d_idx, g['per1'], 5, 1.0), g['err1']):.2e}",
      "| K=41 grid", f"{rel(_hip.fit_errors(y3, d_idx3, g['per3'], 20, 1.0), g['err3']):.2e}",
      "| K=41 lambda 0", f"{rel(_hip.fit_errors(y3, d_idx3, g['per3'], 20, 0.0), g['err3_l0']):.2e}")
