"""Round-2 golden fixtures from the UNMODIFIED reference (build container only).

    python tests/golden/make_golden_r2.py [--only NAME] [--workers 6]

Same import recipe as ``make_golden.py`` (serial ``pqdm`` stand-in).  The
inputs come from ``pyparrm_amd.synth.synth_recording_exact`` -- a generator made of exactly
rounded operations only, so a fixture stores the generator's arguments and the reference's
results, never the multi-megabyte recording, and regenerates bit-identically on the GPU box.

Fixtures written (``r2_*.npz`` / ``r2_periods.json``):

* ``grid26``      26 ``assumed_periods`` (the "1e4 grid" of BASELINE configs[2]) on 2 ch x 10 M and
                  3 ch x 4 M recordings: the reference's period.
* ``cfg1_flow``   BASELINE configs[0]: 1 ch x 60 s @ 22 kHz, ``find_period`` -> ``create_filter`` ->
                  ``filter_data`` (examples/plot_use_parrm.py:77-141 sequence); period, filter taps
                  and a strided sample of the filtered recording.
* ``short``       recordings shorter than the stage lengths (parrm.py:288-301 collapse), and the
                  per-site flow of examples/plot_example_dbs_data.py:52-98 on ``ecog_lfp_data``.
* ``float32``     float32 recordings (the reference standardises in float32, parrm.py:272-280).
* ``fuzz``        60 small recordings over sampling/artefact frequency pairs, channel counts,
                  lengths, detunings and seeds: the reference's period of each.
* ``args``        round 4: 24 recordings with the ARGUMENTS of ``find_period`` varied (parrm.py:148-155): a
                  ``search_samples`` window (float ``arange`` as the reference's tests pass it), two or three
                  ``assumed_periods``, ``outlier_boundary`` 1.5 ... 6, 4-8 channels, 30 000 ... 120 000 samples (stage 3
                  draws its indices at random there) -- written to ``r4_periods.json``.
* ``psd``         the reference's ``compute_psd`` (_utils/_power.py:10-68) on three seeded inputs:
                  ``n_points`` below, equal to and above the recording's length (truncation and
                  zero padding), with and without ``max_freq``, float32 and float64 recordings.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import _import_reference  # noqa: E402
from pyparrm_amd.synth import synth_recording_exact  # noqa: E402

FS, F_ART = 22000.0, 130.0


def assumed_periods_1e4():
    base = FS / F_ART
    return tuple(base * (1 + 0.02 * k) for k in range(-13, 13))


def fuzz_cases():
    """60 deterministic parameter sets (pure Python arithmetic on a seeded integer stream)."""
    rng = np.random.default_rng(20260101)
    pairs = [(22000.0, 130.0), (1000.0, 130.0), (250.0, 130.0), (200.0, 150.0), (4000.0, 130.0),
             (8000.0, 125.0), (2048.0, 140.0), (5000.0, 60.0), (30000.0, 185.0), (512.0, 100.0)]
    cases = []
    for i in range(60):
        fs, fa = pairs[i % len(pairs)]
        n_chans = int(rng.integers(1, 4))
        n_samples = int(rng.choice([900, 2500, 4800, 7000, 12000, 19130, 26000, 40000]))
        detune = float(rng.uniform(-2e-3, 2e-3))
        seed = int(rng.integers(1, 1 << 30))
        random_seed = int(rng.integers(0, 1000))
        gain_hi = float(rng.choice([1.5, 4.0, 10.0]))
        cases.append(dict(tag=f"fuzz{i:02d}", fs=fs, fa=fa, n_chans=n_chans, n_samples=n_samples,
                          period=fs / fa * (1 + detune), seed=seed, random_seed=random_seed,
                          gain=(0.5 * gain_hi, gain_hi), dtype="float64"))
    return cases


def short_cases():
    out = []
    for i, (n_samples, fs, fa) in enumerate([(3000, 1000.0, 130.0), (4999, 22000.0, 130.0),
                                              (7000, 1000.0, 130.0), (9999, 4000.0, 130.0),
                                              (600, 250.0, 130.0)]):
        out.append(dict(tag=f"short{i}", fs=fs, fa=fa, n_chans=2, n_samples=n_samples,
                        period=fs / fa * (1 + 4e-4), seed=500 + i, random_seed=5 + i,
                        gain=(3.0, 8.0), dtype="float64"))
    return out


def float32_cases():
    out = []
    for i, (n_samples, fs, fa) in enumerate([(60000, 1000.0, 130.0), (30000, 22000.0, 130.0),
                                              (26000, 250.0, 130.0)]):
        out.append(dict(tag=f"f32_{i}", fs=fs, fa=fa, n_chans=2, n_samples=n_samples,
                        period=fs / fa * (1 - 3e-4), seed=900 + i, random_seed=11 + i,
                        gain=(3.0, 8.0), dtype="float32"))
    return out


def args_cases():
    """24 deterministic parameter sets with find_period's arguments varied."""
    rng = np.random.default_rng(20261005)
    pairs = [(22000.0, 130.0), (1000.0, 130.0), (4000.0, 130.0), (8000.0, 125.0), (2048.0, 140.0), (30000.0, 185.0)]
    cases = []
    for i in range(24):
        fs, fa = pairs[i % len(pairs)]
        n_samples = int(rng.choice([30000, 45000, 60001, 90000, 120000]))
        case = dict(tag=f"args{i:02d}", fs=fs, fa=fa, n_chans=int(rng.integers(1, 9)), n_samples=n_samples,
                    period=fs / fa * (1 + float(rng.uniform(-2e-3, 2e-3))), seed=int(rng.integers(1, 1 << 30)),
                    random_seed=int(rng.integers(0, 1000)), gain=(2.0, float(rng.choice([4.0, 10.0]))), dtype="float64")
        kind = i % 4
        if kind in (0, 3):  # a window of the recording, as a float arange (tests/test_parrm.py:33-36 passes floats)
            lo = int(rng.integers(0, n_samples // 3))
            hi = int(rng.integers(n_samples // 2, n_samples))
            case["search"] = [lo, hi]
        if kind in (1, 3):  # several assumed periods around the nominal one
            base = fs / fa
            case["assumed"] = [base * (1 + d) for d in rng.uniform(-4e-3, 4e-3, int(rng.integers(2, 4)))]
        if kind in (2, 3):
            case["outlier"] = float(rng.choice([1.5, 2.0, 4.5, 6.0]))
        cases.append(case)
    return cases


def _run_case(case):
    PARRM, _ = _import_reference()
    x = synth_recording_exact(case["n_chans"], case["n_samples"], case["period"], case["seed"],
                              gain_range=case["gain"], dtype=np.dtype(case["dtype"]))
    t0 = time.time()
    p = PARRM(x, case["fs"], case["fa"], verbose=False)
    kw = {}
    if "search" in case:
        kw["search_samples"] = np.arange(float(case["search"][0]), float(case["search"][1]))
    if "outlier" in case:
        kw["outlier_boundary"] = case["outlier"]
    if "assumed" in case:
        kw["assumed_periods"] = tuple(case["assumed"])
    try:
        p.find_period(random_seed=case["random_seed"], **kw)
        period = float(p.period)
    except ValueError as exc:  # "The period cannot be estimated..."
        period = None
        print(case["tag"], "reference raised", exc, flush=True)
    res = dict(case)
    res["ref_period"] = period
    res["ref_seconds"] = round(time.time() - t0, 1)
    print(case["tag"], repr(period), f"{res['ref_seconds']}s", flush=True)
    return res


def run_period_sets(workers, only):
    sets = {"fuzz": fuzz_cases(), "short": short_cases(), "float32": float32_cases()}
    path = os.path.join(HERE, "r2_periods.json")
    done = json.load(open(path)) if os.path.exists(path) else {}
    for name, cases in sets.items():
        if only and only != name:
            continue
        with Pool(workers) as pool:
            done[name] = pool.map(_run_case, cases, chunksize=1)
        json.dump(done, open(path, "w"), indent=1)


def run_args(workers):
    path = os.path.join(HERE, "r4_periods.json")
    with Pool(workers) as pool:
        done = {"args": pool.map(_run_case, args_cases(), chunksize=1)}
    json.dump(done, open(path, "w"), indent=1)


def run_grid26(only):
    path = os.path.join(HERE, "r2_periods.json")
    done = json.load(open(path)) if os.path.exists(path) else {}
    cases = [
        dict(tag="grid26_2x10M", fs=FS, fa=F_ART, n_chans=2, n_samples=10_000_000,
             period=FS / F_ART * (1 + 3e-5), seed=4100, random_seed=44, gain=(2.0, 10.0),
             dtype="float64", assumed=list(assumed_periods_1e4())),
        dict(tag="grid26_3x4M", fs=FS, fa=F_ART, n_chans=3, n_samples=4_000_000,
             period=FS / F_ART * (1 - 2.1e-4), seed=4200, random_seed=7, gain=(1.0, 4.0),
             dtype="float64", assumed=list(assumed_periods_1e4())),
    ]
    with Pool(2) as pool:
        done["grid26"] = pool.map(_run_case, cases, chunksize=1)
    json.dump(done, open(path, "w"), indent=1)


def run_cfg1_flow():
    """examples/plot_use_parrm.py:77-141 on BASELINE configs[0]'s shape."""
    PARRM, paths = _import_reference()
    period_true = FS / F_ART * (1 + 3e-5)
    x = synth_recording_exact(1, 1_320_000, period_true, 3100)
    p = PARRM(x, FS, F_ART, verbose=False)
    t0 = time.time()
    p.find_period(random_seed=44)
    t1 = time.time()
    out = {"n_samples": 1_320_000, "seed": 3100, "true_period": period_true, "random_seed": 44,
           "period": np.float64(p.period)}
    p.create_filter()
    y = p.filter_data()
    t2 = time.time()
    out["default_filter_taps"] = np.flatnonzero(p.filter).astype(np.int32)
    out["default_half_width"] = p.settings["filter"]["filter_half_width"]
    out["default_y_strided"] = y[0, ::997].copy()
    out["default_y_head"] = y[0, :6000].copy()
    out["default_y_tail"] = y[0, -6000:].copy()
    # explicit settings in the style of plot_use_parrm.py:135-140, scaled to this period
    p.create_filter(filter_half_width=4000, omit_n_samples=20, filter_direction="both",
                    period_half_width=1.0)
    y = p.filter_data()
    out["explicit_phw"] = p.settings["filter"]["period_half_width"]
    out["explicit_filter_taps"] = np.flatnonzero(p.filter).astype(np.int32)
    out["explicit_y_strided"] = y[0, ::997].copy()
    out["explicit_y_head"] = y[0, :6000].copy()
    out["explicit_y_tail"] = y[0, -6000:].copy()
    np.savez_compressed(os.path.join(HERE, "r2_cfg1_flow.npz"), **out)
    print("cfg1_flow period", repr(p.period), f"find {t1 - t0:.1f}s filter {t2 - t1:.1f}s", flush=True)

    # per-site flow (plot_example_dbs_data.py:52-98) with a seed so that stage 3 is reproducible
    ecog = np.load(paths("ecog_lfp_data"))
    per = {}
    for name, rows in (("both", [0, 1]), ("ecog", [0]), ("lfp", [1])):
        q = PARRM(ecog[rows], 1000, 130, verbose=False)
        q.find_period(random_seed=44)
        per[name] = np.float64(q.period)
        print("per-site", name, repr(q.period), flush=True)
    np.savez(os.path.join(HERE, "r2_per_site.npz"), **per)


def psd_cases():
    """Inputs of the ``psd`` fixture; a recording is ``synth_recording_exact(n_chans, n_times, period, seed)``."""
    return [
        # n_points < n_times: the first n_points samples only (scipy's fft(x, n) truncates)
        dict(tag="trunc", n_chans=3, n_times=6000, fs=1000, n_points=2000, max_freq=None, dtype="float64",
             period=7.6923, seed=71),
        # n_points > n_times: zero padded; cut at max_freq
        dict(tag="pad_maxfreq", n_chans=2, n_times=1500, fs=22000, n_points=4096, max_freq=5000.0, dtype="float64",
             period=169.23, seed=72),
        # odd n_points == n_times, a float32 recording, cut at max_freq
        dict(tag="odd_f32", n_chans=4, n_times=3001, fs=250, n_points=3001, max_freq=100.0, dtype="float32",
             period=1.923, seed=73),
    ]


def run_psd():
    """The reference's own ``compute_psd`` (``/root/reference/src/pyparrm/_utils/_power.py:10-68``), unmodified."""
    _import_reference()
    from pyparrm._utils._power import compute_psd

    out = {"cases": json.dumps(psd_cases())}
    for case in psd_cases():
        x = synth_recording_exact(case["n_chans"], case["n_times"], case["period"], case["seed"],
                                  dtype=np.dtype(case["dtype"]))
        freqs, psd = compute_psd(x, case["fs"], case["n_points"], case["max_freq"])
        assert psd.dtype == np.float32
        out[f"{case['tag']}_freqs"] = freqs
        out[f"{case['tag']}_psd"] = psd
        print("psd", case["tag"], freqs.shape, psd.shape, float(psd.max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "r2_psd.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--workers", type=int, default=6)
    args = ap.parse_args()
    if args.only in (None, "psd"):
        run_psd()
    if args.only in (None, "args"):
        run_args(args.workers)
    if args.only in (None, "grid26"):
        run_grid26(args.only)
    if args.only in (None, "cfg1_flow"):
        run_cfg1_flow()
    if args.only in (None, "fuzz", "short", "float32"):
        run_period_sets(args.workers, args.only)


if __name__ == "__main__":
    main()
