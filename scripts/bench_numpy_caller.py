#!/usr/bin/env python3
"""NumPy in -> NumPy out `filter_data` on a large host array (the reference's calling convention): serial upload ->
kernel -> read-back against the overlapped channel-block pipeline, and bit-equality of the two."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyparrm_amd import PARRM, parrm as facade

def main():
    c, n = 256, 2_000_000
    rng = np.random.default_rng(0)
    x = rng.standard_normal((c, n))
    p = PARRM(x, 22000.0, 130.0, verbose=False)
    p._period = np.float64(22000.0 / 130.0 * (1 + 3e-5))
    p.create_filter()
    res = {}
    for name, thr in (("serial", 1 << 60), ("pipelined", 256 << 20)):
        facade._PIPELINE_BYTES = thr
        p.filter_data()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); y = p.filter_data(); ts.append(time.perf_counter() - t0)
        res[name] = y.copy()
        print(f"{name:10s} {x.nbytes / 1e9:.1f} GB in, {y.nbytes / 1e9:.1f} GB out: best {min(ts) * 1e3:.0f} ms, median {np.median(ts) * 1e3:.0f} ms", flush=True)
    print("bit-identical:", np.array_equal(res["serial"], res["pipelined"]))

if __name__ == "__main__":
    main()
