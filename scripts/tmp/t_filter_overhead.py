import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pyparrm_amd import PARRM, _hip
from pyparrm_amd.synth import synth_recording_device
x = synth_recording_device(256, 10_000_000, 22000.0, 130.0, seed=0)
torch.cuda.synchronize()
for rep in range(3):
    p = PARRM(x, 22000.0, 130.0, verbose=False)
    p._period = np.float64(169.2358)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p.create_filter()
    t1 = time.perf_counter()
    plan = _hip.FilterPlan(p.filter)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    y = torch.empty_like(x)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    plan.apply(x, out=y)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    y2 = p.filter_data()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"create_filter {1e3*(t1-t0):.3f} plan {1e3*(t2-t1):.3f} empty {1e3*(t3-t2):.3f} apply {1e3*(t4-t3):.3f} filter_data() {1e3*(t5-t4):.3f}")
    del y, y2, p, plan
