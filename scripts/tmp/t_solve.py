import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pyparrm_amd import _hip
_hip.require_gpu()
n, C, P, bw = 24963, 256, 12, 20
y = torch.randn((n, C), dtype=torch.float64, device='cuda')
idx = torch.arange(n, dtype=torch.int64, device='cuda') * 3
per = np.linspace(169.0, 169.5, P)
ws = _hip.FitWorkspace()
for _ in range(20): _hip.fit_errors(y, idx, per, bw, 1.0, ws)
torch.cuda.synchronize()
