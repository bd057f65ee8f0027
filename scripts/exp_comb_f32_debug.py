#!/usr/bin/env python3
"""Where does the float32 variant of the generated kernel differ from the closed form?  (self-test off)"""
import os, sys
import numpy as np
os.environ["PARRM_COMB"] = "force"
os.environ["PARRM_COMB_NO_SELFTEST"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import parrm_oracle as orc
from pyparrm_amd import _hip
_hip.require_gpu()
period = 22000 / 130 * (1 + 3e-5)
n = 300_017
hw = orc.default_filter_half_width(n, period, 0, period / 50)
filt = orc.generate_filter(period, hw, 0, "both", None)
rng = np.random.default_rng(1)
x = np.tile(np.arange(n, dtype=np.float32), (2, 1)) if os.environ.get('RAMP') else rng.standard_normal((2, n)).astype(np.float32)
ref = orc.filter_data_direct(x.astype(np.float64), filt)
plan = _hip.FilterPlan(filt)
y = plan.apply(torch.from_numpy(x).cuda(), out_dtype=torch.float64).cpu().numpy()
print("generated:", plan.generated)
err = np.abs(y - ref)
bad = np.argwhere(err > 1e-9)
print("bad outputs:", bad.shape[0], "of", y.size, "max err", err.max())
np.set_printoptions(precision=4, linewidth=200)
for lo in (0, 3000, 150000, n - 8):
    print("n", lo, "y  ", y[0, lo:lo + 8]); print("      ref", ref[0, lo:lo + 8]); print("      x  ", x[0, lo:lo + 8])
xs = x[0].astype(np.float64)
# is y the filter of some other view of the data?
for name, alt in (("x as pairs swapped", xs.reshape(-1)[: n - n % 2].reshape(-1, 2)[:, ::-1].reshape(-1)),):
    pass
print("corr(y, ref)", np.corrcoef(y[0, 5000:-5000], ref[0, 5000:-5000])[0, 1], "corr(y, x)", np.corrcoef(y[0, 5000:-5000], xs[5000:-5000])[0, 1])
