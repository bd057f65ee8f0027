import itertools, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PARRM_COMB"] = "force"
import torch
from oracle import parrm_oracle as orc
from pyparrm_amd import _hip
x = torch.randn((2, 200_000), dtype=torch.float64, device="cuda")
for period, div in itertools.product([101.77, 123.08, 138.5, 169.2359, 169.5, 175.9], [8, 12, 20]):
    row = []
    for direction, omit, hw in (("both", 0, 2372), ("past", 7, 2372), ("future", 0, 650), ("both", 0, 650)):
        taps = orc.generate_filter(period, hw, omit, direction, period / div)
        plan = _hip.FilterPlan(taps); plan.apply(x)
        st, stride, msg = plan.generated
        row.append((direction, hw, st, stride, msg[:25]))
    print(period, div, row)
