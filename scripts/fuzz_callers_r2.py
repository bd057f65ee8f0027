#!/usr/bin/env python3
"""Randomised checks of the callers either side of the hot path, on one GPU:

A. ``OnlineFilter``: blocks of random length (0 and 1 included), any direction / dtype / output dtype, NaN and
   Inf samples scattered in -- the concatenation must equal the closed form (oracle), zeros in its places;
B. ``filter_file``: ``.npy`` in, ``.npy`` out with random chunk lengths -- the same check;
C. ``ShardedPARRM`` over ``ThreadExchange``: a random recording cut into random channel blocks (2-5 ranks) must
   give, bit for bit, the period, the stage errors and the filtered rows of one ``PARRM`` on the whole;
D. ``find_period_batched``: random groups of recordings -- every period bit-identical to its own
   ``find_period()``.

    python scripts/fuzz_callers_r2.py --cases 40 --seed 0
"""
import argparse
import os
import sys
import tempfile
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--parts", default="ABCD")
    args = ap.parse_args()
    import torch

    from oracle import parrm_oracle as orc
    from pyparrm_amd import PARRM, _hip, find_period_batched
    from pyparrm_amd import sharding as sh
    from pyparrm_amd.streaming import OnlineFilter, filter_file
    from pyparrm_amd.synth import synth_recording_exact

    _hip.require_gpu()
    rng = np.random.default_rng(args.seed)

    def fail(msg):
        print("FAIL", msg, flush=True)
        sys.exit(1)

    def random_filter(n_samples):
        while True:
            period = float(rng.choice([7.7424, 13.0, 64.3, 169.2359, 250.01, 333.3])) * (1 + rng.uniform(-2e-3, 2e-3))
            hw = int(min(max(1, (n_samples - 1) // 2), rng.choice([3, 50, 650, 2372])))
            omit = int(rng.integers(0, max(1, min(hw, 30))))
            direction = str(rng.choice(["both", "past", "future"]))
            phw = float(period / rng.choice([50, 20, 8]))
            try:
                return orc.generate_filter(period, hw, omit, direction, phw), (period, hw, omit, direction, phw)
            except RuntimeError:
                continue

    def random_recording(n_taps_cap=1.0e9):
        n_chans = int(rng.integers(1, 6))
        n_samples = int(rng.choice([17, 500, 5000, 40_000, 131_073, 300_000]))
        filt, fdesc = random_filter(n_samples)
        n_taps = int(np.count_nonzero(filt)) - 1
        while n_taps * n_chans * n_samples > n_taps_cap and n_chans > 1:
            n_chans -= 1
        f32 = bool(rng.random() < 0.4)
        x = rng.standard_normal((n_chans, n_samples)).astype(np.float32 if f32 else np.float64)
        bad = []
        if rng.random() < 0.35 and n_samples >= 500:
            for _ in range(int(rng.integers(1, 4))):
                c, n = int(rng.integers(0, n_chans)), int(rng.integers(0, n_samples))
                x[c, n] = rng.choice([np.nan, np.inf, -np.inf])
                bad.append((c, n))
        return x, filt, fdesc, bad

    def compare(what, y, ref, f32, desc):
        tol = 1e-5 if f32 else 1e-10
        scale = max(float(np.abs(ref).max()), 1e-300)
        if y.shape != ref.shape:
            fail(f"{what}: shape {y.shape} != {ref.shape}: {desc}")
        err = float(np.abs(y.astype(np.float64) - ref).max()) / scale if y.size else 0.0
        if not err <= tol:
            fail(f"{what}: rel err {err:.3e}: {desc}")
        if not np.array_equal(y == 0, ref == 0):
            fail(f"{what}: zero pattern differs in {int(((y == 0) != (ref == 0)).sum())} outputs: {desc}")
        return err

    counts = {"online": 0, "file": 0, "sharded": 0, "batched": 0}
    for case in range(args.cases):
        # ---- A: online
        if "A" in args.parts:
            x, filt, fdesc, bad = random_recording()
            f32 = x.dtype == np.float32
            out_dtype = np.float32 if (f32 and rng.random() < 0.5) else np.float64
            n = x.shape[1]
            cuts = [0]
            while cuts[-1] < n:
                step = int(rng.choice([0, 1, 7, 300, 5000, 70_000]))
                cuts.append(min(n, cuts[-1] + step))
                if len(cuts) > 400:
                    cuts.append(n)
            desc = f"case {case} online: filter {fdesc} x {x.shape} {x.dtype} -> {np.dtype(out_dtype)} bad {bad} blocks {len(cuts) - 1}"
            stream = OnlineFilter(filt, x.shape[0], dtype=x.dtype, out_dtype=out_dtype)
            as_tensor = bool(rng.random() < 0.3)
            parts = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                block = x[:, lo:hi]
                got = stream.push(torch.from_numpy(np.ascontiguousarray(block)).cuda() if as_tensor else block)
                parts.append(got.cpu().numpy() if as_tensor else got)
            parts.append(stream.finish())
            y = np.concatenate(parts, axis=1)
            compare("online", y, orc.filter_data_direct(x.astype(np.float64), filt), f32, desc)
            counts["online"] += 1
        # ---- B: file to file
        if "B" in args.parts and case % 2 == 0:
            x, filt, fdesc, bad = random_recording()
            f32 = x.dtype == np.float32
            out_dtype = np.float32 if (f32 and rng.random() < 0.5) else np.float64
            chunk = int(rng.choice([0, 1, 1000, x.shape[1] // 3 + 1, x.shape[1]]))
            desc = f"case {case} file: filter {fdesc} x {x.shape} {x.dtype} -> {np.dtype(out_dtype)} bad {bad} chunk {chunk}"
            with tempfile.TemporaryDirectory() as tmp:
                src, dst = os.path.join(tmp, "in.npy"), os.path.join(tmp, "out.npy")
                np.save(src, x)
                if chunk == 1 and x.shape[1] > 5000:
                    chunk = 999
                out = filter_file(filt, src, dst, chunk_samples=chunk, out_dtype=out_dtype)
                y = np.array(np.load(dst))
                del out
            compare("file", y, orc.filter_data_direct(x.astype(np.float64), filt), f32, desc)
            counts["file"] += 1
        # ---- C: channel blocks over threads
        if "C" in args.parts and case % 2 == 1:
            world = int(rng.integers(2, 6))
            n_chans = int(rng.integers(world, world + 5))
            n_samples = int(rng.choice([30_000, 80_000, 200_000]))
            fs, fa = float(rng.choice([1000.0, 22000.0])), 130.0
            period = fs / fa * (1 + rng.uniform(-3e-4, 3e-4))
            x = synth_recording_exact(n_chans, n_samples, period, seed=int(rng.integers(1, 10_000)))
            seed = int(rng.integers(0, 1000))
            cuts = sorted(rng.choice(np.arange(1, n_chans), size=world - 1, replace=False).tolist())
            bounds = [0] + cuts + [n_chans]
            desc = f"case {case} sharded: {n_chans} ch x {n_samples} fs {fs} blocks {bounds} seed {seed}"
            whole = PARRM(x, fs, fa, verbose=False)
            whole.find_period(random_seed=seed)
            whole.create_filter()
            y_whole = whole.filter_data()
            results, errors = [None] * world, []

            def rank_main(ex):
                try:
                    torch.cuda.set_device(0)
                    rows = x[bounds[ex.rank]:bounds[ex.rank + 1]]
                    p = sh.ShardedPARRM(rows, fs, fa, ex, verbose=False)
                    p.find_period(random_seed=seed)
                    p.create_filter()
                    results[ex.rank] = (p.period, [t["errors"] for t in p._trace[:3]], p.filter_data())
                except Exception as exc:
                    errors.append(exc)
                    ex._barrier.abort()

            threads = [threading.Thread(target=rank_main, args=(ex,)) for ex in sh.ThreadExchange.group(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                fail(f"{desc}: {errors!r}")
            for rank, (per, stage_errors, _) in enumerate(results):
                if per != whole.period:
                    fail(f"{desc}: rank {rank} period {per!r} != {whole.period!r}")
                for got, want in zip(stage_errors, [t["errors"] for t in whole._trace[:3]]):
                    if not np.array_equal(got, want):
                        fail(f"{desc}: rank {rank} stage errors differ")
            if not np.array_equal(np.concatenate([r[2] for r in results]), y_whole):
                fail(f"{desc}: filtered blocks differ from the whole recording's rows")
            counts["sharded"] += 1
        # ---- D: batched searches
        if "D" in args.parts and case % 3 == 0:
            fs = float(rng.choice([1000.0, 22000.0]))
            group = []
            for _ in range(int(rng.integers(2, 7))):
                n_chans = int(rng.integers(1, 4))
                n_samples = int(rng.choice([3000, 12_000, 40_000, 90_000]))
                period = fs / 130.0 * (1 + rng.uniform(-3e-4, 3e-4))
                group.append(synth_recording_exact(n_chans, n_samples, period, seed=int(rng.integers(1, 10_000))))
            seed = int(rng.integers(0, 1000))
            desc = f"case {case} batched: fs {fs} shapes {[g.shape for g in group]} seed {seed}"
            singles = []
            for g in group:
                p = PARRM(g, fs, 130.0, verbose=False)
                p.find_period(random_seed=seed)
                singles.append(p.period)
            batch = [PARRM(g, fs, 130.0, verbose=False) for g in group]
            find_period_batched(batch, random_seed=seed)
            got = [p.period for p in batch]
            if got != singles:
                fail(f"{desc}: batched {got} != single {singles}")
            counts["batched"] += 1
        if case % 5 == 4:
            print(f"  ... {case + 1} cases {counts}", flush=True)
    torch.cuda.synchronize()
    print(f"{args.cases} cases ok: {counts}")


if __name__ == "__main__":
    main()
