import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from pyparrm_amd import PARRM, _hip
x = torch.randn((256, 10_000_000), dtype=torch.float32, device="cuda")
y = torch.empty((256, 10_000_000), dtype=torch.float64, device="cuda")
p = PARRM(np.zeros((1, 10_000_000)), 22000.0, 130.0, verbose=False); p._period = np.float64(22000.0/130.0*(1+3e-5)); p.create_filter()
plans = {}
plan = _hip.FilterPlan(p.filter)
plans = {"packed reads, f64 sums (default)": {}}
res = {k: [] for k in plans}
for rnd in range(8):
    for k, env in plans.items():
        os.environ.update(env)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record(); plan.apply(x,out=y); e1.record(); torch.cuda.synchronize()
        for name in env: os.environ.pop(name)
        if rnd: res[k].append(e0.elapsed_time(e1))
for k, ts in res.items(): print("f32 -> f64, %s: median %.3f ms" % (k, np.median(ts)))
