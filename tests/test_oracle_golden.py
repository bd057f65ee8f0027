"""Pin the CPU oracle (oracle/parrm_oracle.py) against the reference.

Pins: (1) ``matlab_filtered.npy`` -- the reference's only numerical known-answer
(examples/plot_use_parrm.py:77-80,135-141,235-240); (2) fixtures written by
``tests/golden/make_golden.py`` from the unmodified reference.
"""

import os

import numpy as np
import pytest

from oracle import parrm_oracle as orc
from pyparrm_amd import get_example_data_paths

DIRS = {0: "both", 1: "past", 2: "future"}


def test_standardise(golden):
    g = golden("standardise.npz")
    out = orc.standardise_data(g["x"], float(g["outlier_boundary"]))
    assert out.dtype == g["std"].dtype
    np.testing.assert_array_equal(out, g["std"])


def test_fit_errors(golden):
    g = golden("fit_errors_ecog.npz")
    ecog = np.load(get_example_data_paths("ecog_lfp_data"))
    std = orc.standardise_data(ecog, 3.0)
    np.testing.assert_array_equal(std[:, g["idx1"][:64]], g["std_cols1"])
    # same NumPy/LAPACK expressions -> bit-identical here
    e1 = orc.grid_errors(g["per1"], std, g["idx1"], 5, 1.0)
    np.testing.assert_array_equal(e1, g["err1"])
    e3 = orc.grid_errors(g["per3"][:3], std, g["idx3"], 20, 1.0)
    np.testing.assert_array_equal(e3, g["err3"][:3])
    e30 = orc.grid_errors(g["per3"][:3], std, g["idx3"], 20, 0.0)
    np.testing.assert_array_equal(e30, g["err3_l0"][:3])
    ea = orc.fit_error(np.array([7.7424]), std, g["idx1"], 10, 1.0)
    assert float(ea) == float(g["err_arr"])


def test_stage_indices(golden):
    g = golden("fit_errors_ecog.npz")
    rng = np.random.default_rng(44)
    search = np.arange(60001 - 1)
    i1 = orc.centre_indices(search, 60001, 5000, 0.0, rng)
    i2 = orc.centre_indices(search, 60001, 10000, 0.0, rng)
    i3 = orc.centre_indices(search, 60001, 25000, 0.95, rng)
    np.testing.assert_array_equal(i1, g["idx1"])
    assert i2.shape[0] in (10000, 10001)
    np.testing.assert_array_equal(i3, g["idx3"])


def test_possible_periods_shape():
    grid = orc.possible_periods((1000 / 130,), 1)
    assert 380 <= grid.shape[0] <= 402
    assert np.all(np.diff(grid) > 0)
    grid26 = orc.possible_periods(tuple(169.0 * (1 + 0.02 * k) for k in range(-13, 13)), 1)
    assert grid26.shape[0] > 9000


def test_period_example_data(golden):
    g = golden("periods.npz")
    x = np.load(get_example_data_paths("example_data"))
    period = orc.find_period(x, 200, 150)
    assert period == g["example_data"]
    assert period == 1.3311148014466094  # SURVEY.md section 4 / BASELINE.md


def test_period_small_synth(golden):
    g = golden("periods.npz")
    period = orc.find_period(
        g["synth_3x6000"], 1000, 130, search_samples=np.arange(0, 3000.0),
        assumed_periods=(7.6, 7.7), random_seed=1,
    )
    assert period == g["synth_3x6000_half_two_estimates"]


def test_filters(golden):
    g = golden("filters.npz")
    n_samples = int(g["n_samples"])
    for tag in "abcdefgh":
        per, hw, omit, d, phw = g[f"{tag}_params"]
        ref = g[f"{tag}_filter"]
        hw = int(hw)
        assert ref.shape[0] == 2 * hw + 1
        out = orc.generate_filter(per, hw, int(omit), DIRS[int(d)], phw)
        np.testing.assert_array_equal(out, ref)
    # default half-widths reproduce (parrm.py:788-801)
    for tag in "bcde":
        per, hw, omit, d, phw = g[f"{tag}_params"]
        assert orc.default_filter_half_width(n_samples, per, int(omit), phw) == int(hw)
    per, hw, *_ = g["b_params"]
    assert int(hw) == 2372  # 22 kHz / 130 Hz geometry


def test_filter_data_matlab_known_answer(golden):
    """The reference's own numerical pin: PyPARRM output == MATLAB PARRM output."""
    g = golden("filter_data.npz")
    x = np.load(get_example_data_paths("example_data"))
    matlab = np.load(get_example_data_paths("matlab_filtered"))
    filt = orc.generate_filter(1.3311148014466094, 2000, 20, "both", 0.01)
    np.testing.assert_array_equal(filt, g["example_filter"])
    assert int((filt < 0).sum()) == 62
    y_fft = orc.filter_data_fft(x, filt)
    y_dir = orc.filter_data_direct(x, filt)
    np.testing.assert_array_equal(y_fft, g["example_filtered"])
    assert np.allclose(y_fft, matlab)  # the reference's check, plot_use_parrm.py:239
    assert np.abs(y_fft - matlab).max() < 1e-13
    assert np.abs(y_dir - matlab).max() < 1e-13
    assert np.abs(y_dir - y_fft).max() < 1e-13


@pytest.mark.parametrize("tag", ["both", "past", "future"])
def test_filter_data_synth(golden, tag):
    g = golden("filter_data.npz")
    x = g["synth_x"]
    filt = g[f"synth_{tag}_filter"]
    ref = g[f"synth_{tag}_y"]
    np.testing.assert_array_equal(orc.filter_data_fft(x, filt), ref)
    direct = orc.filter_data_direct(x, filt)
    cnt = orc.valid_tap_counts(filt, x.shape[1])
    ok = cnt > 0  # zero-count samples are FFT noise in the reference
    scale = np.abs(ref).max()
    assert np.abs(direct - ref)[:, ok].max() <= 1e-12 * scale
    assert np.all(direct[:, ~ok] == 0)
    # data shorter than the filter (tests/test_parrm.py:58-60)
    xs, ys = g[f"short_{tag}_x"], g[f"short_{tag}_y"]
    np.testing.assert_array_equal(orc.filter_data_fft(xs, filt), ys)
    ds = orc.filter_data_direct(xs, filt)
    oks = orc.valid_tap_counts(filt, xs.shape[1]) > 0
    if oks.any():
        assert np.abs(ds - ys)[:, oks].max() <= 1e-12 * max(np.abs(ys).max(), 1.0)
    assert np.all(ds[:, ~oks] == 0)


def test_valid_tap_counts_bruteforce():
    rng = np.random.default_rng(3)
    for n_samples in (1, 7, 50, 131):
        filt = orc.generate_filter(7.3, 40, 2, "both", 0.6)
        w = np.arange(-40, 41)
        taps = w[(filt != 0) & (w != 0)]
        brute = np.array(
            [sum(1 for t in taps if 0 <= n - t < n_samples) for n in range(n_samples)]
        )
        np.testing.assert_array_equal(orc.valid_tap_counts(filt, n_samples), brute)
    assert rng is not None


def test_filter_data_f32_promotes(golden):
    g = golden("filter_data.npz")
    assert str(g["synth_f32_dtype"]) == "float64"
    filt = g["synth_both_filter"]
    y = orc.filter_data_fft(g["synth_x"].astype(np.float32), filt)
    assert y.dtype == np.float64
    np.testing.assert_array_equal(y, g["synth_f32_y"])


# ---------------------------------------------------------------------------- round-2 fixtures
def _r2():
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r2_periods.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("tag", ["short0", "short1", "short2", "short4", "fuzz00", "fuzz07", "fuzz09", "fuzz37"])
def test_oracle_period_matches_reference_r2(tag):
    """The oracle's whole search against periods written by the unmodified reference
    (tests/golden/make_golden_r2.py), including recordings shorter than the stage lengths
    (parrm.py:288-301) -- bit for bit."""
    from pyparrm_amd.synth import synth_recording_exact

    fixtures = _r2()
    case = next(c for c in fixtures["fuzz"] + fixtures["short"] if c["tag"] == tag)
    x = synth_recording_exact(case["n_chans"], case["n_samples"], case["period"], case["seed"],
                              gain_range=tuple(case["gain"]))
    got = orc.find_period(x, case["fs"], case["fa"], random_seed=case["random_seed"])
    assert float(got) == case["ref_period"]


@pytest.mark.parametrize("tag", ["args20", "args15", "args22"])
def test_oracle_period_matches_reference_r4_arguments(tag):
    """Round-4 fixtures (``make_golden_r2.py --only args``): ``find_period`` with its ARGUMENTS varied -- a float
    ``search_samples`` window (args20), two assumed periods + ``outlier_boundary`` 2.0 + a window (args15),
    ``outlier_boundary`` 6.0 (args22) -- the oracle's whole search against the unmodified reference's period, bit for
    bit (parrm.py:213-270 sorts and validates the window, :272-280 clips at the boundary, :376-405 builds the grid)."""
    import json

    from pyparrm_amd.synth import synth_recording_exact

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_periods.json")) as fh:
        case = next(c for c in json.load(fh)["args"] if c["tag"] == tag)
    x = synth_recording_exact(case["n_chans"], case["n_samples"], case["period"], case["seed"],
                              gain_range=tuple(case["gain"]))
    kw = {}
    if "search" in case:
        kw["search_samples"] = np.arange(float(case["search"][0]), float(case["search"][1]))
    if "assumed" in case:
        kw["assumed_periods"] = tuple(case["assumed"])
    if "outlier" in case:
        kw["outlier_boundary"] = case["outlier"]
    got = orc.find_period(x, case["fs"], case["fa"], random_seed=case["random_seed"], **kw)
    assert float(got) == case["ref_period"]


def test_r2_fixture_inventory():
    fixtures = _r2()
    assert len(fixtures["fuzz"]) == 60 and len(fixtures["short"]) == 5
    assert len(fixtures["grid26"]) == 2 and len(fixtures["float32"]) == 3
    assert all(c["ref_period"] is not None for group in fixtures.values() for c in group)
