#!/usr/bin/env python3
"""Does a big grid finish sooner as candidate slices on two streams (one slice's design-matrix kernel -- HBM-write
bound -- beside the other's Gram kernel -- matrix-core bound)?  Slices are planned as the whole grid, so the errors are
the single call's bits either way."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import _hip

C = 256
L = _hip.lib()
g = torch.Generator(device="cuda").manual_seed(1)
streams = [torch.cuda.Stream() for _ in range(2)]
for P, n, bw in [(10044, 5001, 5), (381, 24963, 20)]:
    y = torch.randn((n, C), generator=g, device="cuda", dtype=torch.float64).clamp_(-3, 3)
    idx = torch.arange(1000, 1000 + n, device="cuda", dtype=torch.int64)
    periods = 169.2 * (1 + np.linspace(-1e-2, 1e-2, P))
    ws0 = _hip.FitWorkspace()
    whole = _hip.fit_errors(y, idx, periods, bw, 1.0, ws0)
    d_per = torch.from_numpy(periods).cuda()
    d_err = torch.zeros(P, device="cuda", dtype=torch.float64)

    def run(chunks, n_streams):
        cuts = np.linspace(0, P, chunks + 1).astype(int)
        bufs = []
        for i in range(chunks):
            cnt = int(cuts[i + 1] - cuts[i])
            nbytes = L.parrm_fit_slice_workspace_bytes(n, C, cnt, P, bw)
            bufs.append((torch.empty(nbytes, dtype=torch.uint8, device="cuda"), nbytes))

        def once():
            start = torch.cuda.Event()
            start.record()
            for st in streams[:n_streams]:
                st.wait_event(start)
            for i in range(chunks):
                lo, cnt = int(cuts[i]), int(cuts[i + 1] - cuts[i])
                st = streams[i % n_streams]
                _hip.check(L.parrm_fit_errors_slice(y.data_ptr(), y.stride(0), idx.data_ptr(), n, C, d_per.data_ptr() + 8 * lo, cnt,
                                                    P, bw, 1.0, d_err.data_ptr() + 8 * lo, bufs[i][0].data_ptr(), bufs[i][1],
                                                    int(st.cuda_stream)), "slice")
            for st in streams[:n_streams]:
                torch.cuda.current_stream().wait_stream(st)

        for _ in range(2):
            once()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            once()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        same = bool(np.array_equal(d_err.cpu().numpy(), whole))
        return best, same

    for chunks, ns in [(1, 1), (2, 1), (2, 2), (4, 2), (8, 2)]:
        ms, same = run(chunks, ns)
        print(f"P={P} bw={bw}: {chunks} slices on {ns} stream(s): {ms:.3f} ms   bits equal to one call: {same}", flush=True)
