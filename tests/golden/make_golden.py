"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What it does: puts ``/root/reference/src`` on ``sys.path``, imports ``pyparrm`` and calls
its private hot-path methods on seeded inputs, saving inputs + outputs as ``.npz`` (data
only; no reference source or bytecode is written anywhere).

The reference imports ``pqdm.threads.pqdm``, which is not installed in this image and
cannot be installed offline.  ``pqdm`` does no arithmetic: it is an ordered thread-pool
``map`` with a progress bar (parrm.py:445-454, :510-517).  This script registers an
in-process module of that name whose ``pqdm`` applies the function serially and returns
the ordered list, which is exactly ``pqdm(..., n_jobs=1)``'s result.
"""

from __future__ import annotations

import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    shim = types.ModuleType("pqdm")
    threads = types.ModuleType("pqdm.threads")

    def pqdm(args, fn, n_jobs, argument_type=None, **_kw):
        assert argument_type == "kwargs"
        return [fn(**a) for a in args]

    threads.pqdm = pqdm
    shim.threads = threads
    sys.modules["pqdm"] = shim
    sys.modules["pqdm.threads"] = threads
    sys.path.insert(0, "/root/reference/src")
    import pyparrm  # noqa: F401
    from pyparrm import PARRM, get_example_data_paths

    return PARRM, get_example_data_paths


def synth(n_chans, n_samples, fs, f_art, seed, detune=3e-5):
    """Small synthetic recording: white noise + 20-harmonic periodic artefact."""
    period = fs / f_art * (1 + detune)
    rng = np.random.default_rng(seed)
    amp = 1.0 / np.arange(1, 21)
    ph = rng.uniform(0, 2 * np.pi, 20)
    gain = rng.uniform(5, 20, n_chans)
    off = rng.uniform(0, period, n_chans)
    n = np.arange(n_samples)
    x = rng.standard_normal((n_chans, n_samples))
    for c in range(n_chans):
        th = 2 * np.pi * (n + off[c]) / period
        art = sum(amp[k] * np.sin((k + 1) * th + ph[k]) for k in range(20))
        x[c] += gain[c] * art
    return x


def main():
    PARRM, paths = _import_reference()
    out = {}

    # ---------------------------------------------------------------- (i) standardise
    rng = np.random.default_rng(101)
    x = rng.standard_normal((3, 257)) * np.array([[1.0], [10.0], [0.01]]) + 5.0
    x[1, 40] = 300.0  # outlier -> clipped
    p = PARRM(x, 100, 13, verbose=False)
    p._outlier_boundary = 3.0
    p._standardise_data()
    np.savez(os.path.join(HERE, "standardise.npz"), x=x, outlier_boundary=3.0,
             std=p._standard_data)

    # ---------------------------------------------------------------- (ii) fit errors
    ecog = np.load(paths("ecog_lfp_data"))
    p = PARRM(ecog, 1000, 130, verbose=False)
    p._check_sort_find_stim_period_inputs(None, None, 3.0, 44, 1)
    p._standardise_data()
    rs = np.random.default_rng(44)
    cases = {}
    # stage-1 like: contiguous centre block, bw 5, lambda 1
    idx1 = p._get_centre_indices(5000, 0.0, rs)
    per1 = (1000 / 130) * (1 + np.linspace(-1e-2, 1e-2, 21))
    err1 = np.array([p._optimise_local(q, p._standard_data, idx1, 5, 1.0) for q in per1])
    # stage-3 like: random unique indices, bw 20, lambda 1 and lambda 0
    _ = p._get_centre_indices(10000, 0.0, rs)
    idx3 = p._get_centre_indices(25000, 0.95, rs)
    per3 = 7.7424 * (1 + np.linspace(-1e-4, 1e-4, 9))
    err3 = np.array([p._optimise_local(q, p._standard_data, idx3, 20, 1.0) for q in per3])
    err3_l0 = np.array([p._optimise_local(q, p._standard_data, idx3, 20, 0.0) for q in per3])
    # period passed as shape-(1,) array, as fmin does
    err_arr = p._optimise_local(np.array([7.7424]), p._standard_data, idx1, 10, 1.0)
    np.savez(os.path.join(HERE, "fit_errors_ecog.npz"), idx1=idx1, per1=per1, err1=err1,
             idx3=idx3, per3=per3, err3=err3, err3_l0=err3_l0, err_arr=np.float64(err_arr),
             std_cols1=p._standard_data[:, idx1[:64]])

    # ---------------------------------------------------------------- (iii) periods
    periods = {}
    ex = np.load(paths("example_data"))
    t0 = time.time()
    p = PARRM(ex, 200, 150, verbose=False)
    p.find_period()
    periods["example_data"] = p.period
    print("example_data period", repr(p.period), f"{time.time() - t0:.1f}s")

    t0 = time.time()
    p = PARRM(ecog, 1000, 130, verbose=False)
    p.find_period(random_seed=44)
    periods["ecog_lfp_data_seed44"] = p.period
    print("ecog period", repr(p.period), f"{time.time() - t0:.1f}s")

    xs = synth(4, 30000, 22000, 130, seed=5)
    t0 = time.time()
    p = PARRM(xs, 22000, 130, verbose=False)
    p.find_period(random_seed=3)
    periods["synth_4x30000_seed3"] = p.period
    print("synth period", repr(p.period), f"{time.time() - t0:.1f}s")

    xs2 = synth(3, 6000, 1000, 130, seed=9)
    half = np.arange(0, 3000.0)
    p = PARRM(xs2, 1000, 130, verbose=False)
    p.find_period(search_samples=half, assumed_periods=(7.6, 7.7), random_seed=1)
    periods["synth_3x6000_half_two_estimates"] = p.period
    print("synth2 period", repr(p.period))

    np.savez(os.path.join(HERE, "periods.npz"), synth_4x30000=xs, synth_3x6000=xs2,
             **{k: np.float64(v) for k, v in periods.items()})

    # ---------------------------------------------------------------- (iv) filters
    filt = {}
    p = PARRM(np.zeros((1, 20001)), 22000, 130, verbose=False)
    settings = [
        # (tag, period, hw, omit, direction, phw)
        ("a", 1.3311148014466094, 2000, 20, "both", 0.01),
        ("b", 169.23584615384616, None, 0, "both", None),
        ("c", 169.23584615384616, None, 0, "past", None),
        ("d", 169.23584615384616, None, 5, "future", None),
        ("e", 7.742402205597892, None, 0, "both", None),
        ("f", 7.742402205597892, 300, 10, "past", 0.5),
        ("g", 2.023966953751087, 49, 0, "both", None),
        ("h", 50.0, 120, 0, "both", 50.0),
    ]
    for tag, per, hw, omit, direction, phw in settings:
        p._period = np.float64(per)
        p._check_sort_create_filter_inputs(hw, omit, direction, phw)
        p._generate_filter()
        filt[f"{tag}_filter"] = p._filter
        filt[f"{tag}_params"] = np.array(
            [per, p._filter_half_width, omit, {"both": 0, "past": 1, "future": 2}[direction],
             p._period_half_width]
        )
    np.savez(os.path.join(HERE, "filters.npz"), n_samples=20001, **filt)

    # ---------------------------------------------------------------- (v) filter_data
    fd = {}
    # example + MATLAB known answer settings (plot_use_parrm.py:135-141)
    p = PARRM(ex, 200, 150, verbose=False)
    p._period = np.float64(periods["example_data"])
    p.create_filter(2000, 20, "both", 0.01)
    fd["example_filtered"] = p.filter_data()
    fd["example_filter"] = p.filter

    rng = np.random.default_rng(7)
    x = synth(3, 6000, 22000, 130, seed=11)
    for tag, direction, omit in (("both", "both", 0), ("past", "past", 0), ("future", "future", 3)):
        p = PARRM(x, 22000, 130, verbose=False)
        p._period = np.float64(169.23584615384616)
        p.create_filter(None, omit, direction, None)
        fd[f"synth_{tag}_filter"] = p.filter
        fd[f"synth_{tag}_y"] = p.filter_data()
        # data shorter than the filter (tests/test_parrm.py:58-60)
        short = rng.standard_normal((2, 50))
        fd[f"short_{tag}_x"] = short
        fd[f"short_{tag}_y"] = p.filter_data(short)
    fd["synth_x"] = x
    # float32 in -> float64 out
    p = PARRM(x, 22000, 130, verbose=False)
    p._period = np.float64(169.23584615384616)
    p.create_filter()
    x32 = x.astype(np.float32)
    y32 = p.filter_data(x32)
    fd["synth_f32_y"] = y32
    fd["synth_f32_dtype"] = np.array(str(y32.dtype))
    np.savez_compressed(os.path.join(HERE, "filter_data.npz"), **fd)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
