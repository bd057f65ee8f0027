"""CPU baseline leg of ``bench.py`` -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

Times the oracle (the reference's own formulation: NumPy/LAPACK harmonic regressions,
``scipy.optimize.fmin``, two ``scipy.signal.convolve`` calls per ``filter_data``) on the host cores
of the box the GPU benchmark runs on, as BASELINE.md section 3 lays out:

* config 1 in full: 1 ch x 1.32 M (60 s @ 22 kHz), ``find_period`` (default grid) + ``filter_data``;
* config 2 in full: 64 ch x 1 M, ``filter_data`` only;
* config 3 on a stated subsample: 8 of the 256 channels x 10 M samples, ``find_period`` over the
  same 26-estimate ("1e4") grid + ``filter_data``.  Both legs cost the same per channel (the
  regression loops over channels, parrm.py:588; the convolution is per column, :861-864), so the
  full configuration is 32x the subsample's time at the same Msamples/s.

Parallelism mirrors what the reference offers: ``find_period(n_jobs=...)`` maps candidates and
Nelder-Mead starts over a pool (``pqdm``, parrm.py:445-454, :510-517); here the pool is processes
(one BLAS thread each) instead of threads, and ``filter_data`` -- single-threaded in the reference --
is additionally spread over channels, which flatters the baseline.  ``cores`` = pool size.

Runs as its own process, before the benchmark touches the GPU:

    python -m oracle.cpu_baseline [--procs N] [--quick]      -> one JSON object on stdout
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

for _var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_var] = "1"  # one core per pool worker: `cores` below is the truth

import multiprocessing as mp  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import parrm_oracle as orc  # noqa: E402
from pyparrm_amd.synth import synth_recording_exact  # noqa: E402

FS, F_ART = 22000.0, 130.0
_TASK = None  # the callable of the map in flight: forked workers inherit it (no pickling of big arrays)


def assumed_periods_1e4():
    base = FS / F_ART
    return tuple(base * (1 + 0.02 * k) for k in range(-13, 13))


def _call_task(item):
    return _TASK(item)


class ForkMapper:
    """Ordered process-pool map.  The pool is forked per call, after the callable (which carries the
    standardised recording) has been parked in a module global, so only the items travel."""

    def __init__(self, procs: int):
        self.procs = procs
        self.ctx = mp.get_context("fork")

    def __call__(self, function, items):
        global _TASK
        items = list(items)
        if self.procs <= 1 or len(items) <= 1:
            return [function(item) for item in items]
        _TASK = function
        try:
            with self.ctx.Pool(min(self.procs, len(items))) as pool:
                chunk = max(1, len(items) // (8 * self.procs))
                return pool.map(_call_task, items, chunksize=chunk)
        finally:
            _TASK = None


def _recording(n_chans, n_samples, seed, mapper):
    """``synth_recording_exact`` rows, generated in the pool (one task per channel)."""
    period = FS / F_ART * (1 + 3e-5)
    rows = mapper(_Row(n_chans, n_samples, period, seed), range(n_chans))
    return np.stack(rows)


class _Row:
    def __init__(self, n_chans, n_samples, period, seed):
        self.args = (n_chans, n_samples, period, seed)

    def __call__(self, chan):
        n_chans, n_samples, period, seed = self.args
        r = np.random.default_rng(seed + 7919)
        gain = r.uniform(2.0, 10.0, n_chans)[chan]
        offset = r.uniform(0.0, period, n_chans)[chan]
        from pyparrm_amd.synth import pulse_shape

        rng = np.random.default_rng(seed + chan)
        out = np.empty(n_samples)
        chunk = 1 << 21
        for lo in range(0, n_samples, chunk):
            hi = min(lo + chunk, n_samples)
            n = np.arange(lo, hi, dtype=np.float64)
            out[lo:hi] = rng.standard_normal(hi - lo) + gain * pulse_shape(np.mod((n + offset) / period, 1.0))
        return out


class _FilterRow:
    def __init__(self, x, filt):
        self.x, self.filt = x, filt

    def __call__(self, chan):
        return orc.filter_data_fft(self.x[chan:chan + 1], self.filt)


def run_config(name, n_chans, n_samples, mapper, find=True, assumed=None, seed=6000, save=None):
    x = _recording(n_chans, n_samples, seed, mapper)
    if save:  # bench.py loads these rows as the first channels of the recording the GPU works on
        np.save(save, x)
    out = {"config": name, "n_chans": n_chans, "n_samples": n_samples}
    t_find = 0.0
    if find:
        t0 = time.perf_counter()
        period = orc.find_period(x, FS, F_ART, assumed_periods=assumed, random_seed=44, mapper=mapper)
        t_find = time.perf_counter() - t0
        period = float(np.asarray(period).reshape(-1)[0])
    else:
        period = FS / F_ART * (1 + 3e-5)
    hw = orc.default_filter_half_width(n_samples, period, 0, period / 50)
    filt = orc.generate_filter(period, hw, 0, "both", None)
    t0 = time.perf_counter()
    rows = mapper(_FilterRow(x, filt), range(n_chans))
    t_filt = time.perf_counter() - t0
    del rows
    total = t_find + t_filt
    out.update(period=period, find_period_s=round(t_find, 2), filter_data_s=round(t_filt, 2),
               msamples_per_s=n_chans * n_samples / total / 1e6)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--quick", action="store_true", help="tiny sizes (tests)")
    ap.add_argument("--save-cfg3", default=None, help="write the configs[2] subsample's channels to this .npy file")
    args = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = args.procs or max(1, min(16, avail))
    mapper = ForkMapper(procs)
    scale = 100 if args.quick else 1
    t0 = time.perf_counter()
    cfg1 = run_config("configs[0]: 1 ch x 60 s @ 22 kHz, find_period (default grid) + filter_data, in full",
                      1, 1_320_000 // scale, mapper, seed=6100)
    cfg2 = run_config("configs[1]: 64 ch x 1 M, filter_data only, in full", 64 // (8 if args.quick else 1),
                      1_000_000 // scale, mapper, find=False, seed=6200)
    cfg3 = run_config("configs[2] subsample: 8 of 256 channels x 10 M, find_period over the 26-estimate grid "
                      "+ filter_data", 8 // (4 if args.quick else 1), 10_000_000 // scale, mapper,
                      assumed=assumed_periods_1e4(), seed=6300, save=args.save_cfg3)
    result = {
        "value": cfg3["msamples_per_s"],
        "unit": "Msamples/s",
        "cores": procs,
        "kind": "port",
        "sample": (
            f"oracle (reference formulation) on {procs} processes x 1 BLAS thread "
            f"({os.cpu_count()} logical CPUs on the host, {avail} usable): value = configs[2] on 8 of its 256 "
            f"channels x {cfg3['n_samples']} samples (find_period over 10 044 + 387 + 381 candidates "
            f"{cfg3['find_period_s']} s, filter_data {cfg3['filter_data_s']} s); both legs cost the same per "
            "channel, so the full 256-channel configuration takes 32x as long at this Msamples/s. Also timed "
            f"in full: configs[0] {cfg1['msamples_per_s']:.3f} Msamples/s (find {cfg1['find_period_s']} s, "
            f"filter {cfg1['filter_data_s']} s); configs[1] filter_data {cfg2['msamples_per_s']:.2f} Msamples/s "
            f"({cfg2['filter_data_s']} s). The reference's filter_data is single-threaded; spreading it over "
            "channels flatters the baseline."
        ),
        "period": cfg3["period"],
        "configs": [cfg1, cfg2, cfg3],
        "wall_s": round(time.perf_counter() - t0, 1),
    }
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
