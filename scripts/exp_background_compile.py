#!/usr/bin/env python3
"""First and second filter_data call for filter geometries no cache has a code object for (background compile), with
the plan's creation timed on its own: which part of a first call is the plan search, which the launch."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARRM_KERNEL_CACHE"] = tempfile.mkdtemp()
import numpy as np, torch
from pyparrm_amd import PARRM, _hip
x = torch.randn((4, 9_000_000), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for fs, fa in ((20000.0, 187.0), (12000.0, 131.0), (16000.0, 129.0), (18000.0, 133.0), (21000.0, 127.0), (20000.0, 187.5)):
    p = PARRM(x, fs, fa, verbose=False)
    p._period = np.float64(fs / fa * (1 + 1e-4))
    p.create_filter()
    t0 = time.perf_counter(); plan = _hip.FilterPlan(p.filter); t1 = time.perf_counter()
    y = p.filter_data(); torch.cuda.synchronize(); t2 = time.perf_counter()
    y = p.filter_data(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(fs, fa, "plan create %.1f ms  first filter %.1f ms  second %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), p._last_plan.generated[:2], "taps", int((p.filter != 0).sum()) - 1, flush=True)
time.sleep(3)
