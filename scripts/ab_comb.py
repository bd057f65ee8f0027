#!/usr/bin/env python3
"""Interleaved A/B of generated-comb-kernel variants on one device: every arm is a set of NAME=VALUE settings in
force while ITS plan is built and launched (generator knobs are read when the kernel is generated, i.e. at the
plan's first large launch).  `phase` as an arm name = the phase-major kernel (PARRM_COMB=0); `copy` = a plain device copy of the recording into the
output (torch's vectorised copy kernel: the same 16 B/sample, the box's copy ceiling in the same process).

    python scripts/ab_comb.py default PARRM_COMB_DEBUG=1 PARRM_COMB_BATCH=5,PARRM_COMB_STRETCH=524288 phase
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("arms", nargs="+")
    ap.add_argument("--chans", type=int, default=256)
    ap.add_argument("--samples", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--fs", type=float, default=22000.0)
    ap.add_argument("--fa", type=float, default=130.0)
    args = ap.parse_args()
    import torch

    from pyparrm_amd import PARRM, _hip

    _hip.require_gpu()
    os.environ["PARRM_COMB_VERBOSE"] = "1"
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((args.chans, args.samples), dtype=torch.float64, device="cuda", generator=g)
    y = torch.empty_like(x)
    p = PARRM(np.zeros((1, args.samples)), args.fs, args.fa, verbose=False)
    p._period = np.float64(args.fs / args.fa * (1 + 3e-5))
    p.create_filter()

    def settings(arm):
        if arm == "default":
            return {"PARRM_COMB": "force"}
        if arm == "phase":
            return {"PARRM_COMB": "0"}
        d = dict(kv.split("=", 1) for kv in arm.split(","))
        d.setdefault("PARRM_COMB", "force")
        return d

    plans = {}
    times = {a: [] for a in args.arms}
    for rnd in range(args.rounds + 1):
        for arm in args.arms:
            if arm == "copy":
                y.copy_(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    y.copy_(x)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[arm].append(e0.elapsed_time(e1) / 3)
                continue
            env = settings(arm)
            os.environ.update(env)
            if arm not in plans:
                plans[arm] = _hip.FilterPlan(p.filter)
            plan = plans[arm]
            plan.apply(x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                plan.apply(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[arm].append(e0.elapsed_time(e1) / 3)
            for k in env:
                os.environ.pop(k, None)
    b = 16 * args.chans * args.samples
    for arm in args.arms:
        t = np.array(times[arm])
        print(f"{arm:60s} {t.min():7.3f} ms min {np.median(t):7.3f} med  {b / t.min() / 1e9:6.3f} TB/s  frac {b / t.min() / 8e9:5.3f}", flush=True)


if __name__ == "__main__":
    main()
