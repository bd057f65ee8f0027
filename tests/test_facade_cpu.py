"""Host-side contract of the ``PARRM`` façade (no GPU needed).

Mirrors the reference's behavioural tests (tests/test_parrm.py:71-346, rewritten against this
package): exact error messages, validation order, the call-order state machine and the
``settings`` layout; plus create_filter parity with the golden filters, and the guarantee that
the device path fails loudly -- never silently falls back -- when no GPU is present.
"""

from multiprocessing import cpu_count

import numpy as np
import pytest

from pyparrm_amd import PARRM, _hip, get_example_data_paths

FS, FA = 20, 10
DIRS = {0: "both", 1: "past", 2: "future"}


def _data(shape=(1, 100)):
    return np.random.default_rng(44).standard_normal(shape)


def _with_period(data, period, fs=FS, fa=FA):
    """A façade whose period has been injected (the way parity tests pin the taps)."""
    p = PARRM(data, fs, fa, verbose=False)
    p._period = np.float64(period)
    return p


def _has_gpu():
    try:
        _hip.require_gpu()
        return True
    except Exception:
        return False


def test_exports_and_datasets():
    import pyparrm_amd

    assert set(pyparrm_amd.__all__) >= {"PARRM", "get_example_data_paths"}
    shapes = {"example_data": (1, 19130), "example_data_artefact_free": (1, 19130),
              "matlab_filtered": (1, 19130), "ecog_lfp_data": (2, 60001)}
    for name, shape in shapes.items():
        assert np.load(get_example_data_paths(name)).shape == shape
    with pytest.raises(ValueError, match="`name` must be one of"):
        get_example_data_paths("nope")


def test_init_wrong_types_and_values():
    d = _data()
    with pytest.raises(TypeError, match="`data` must be a NumPy array."):
        PARRM(data=d.tolist(), sampling_freq=FS, artefact_freq=FA)
    with pytest.raises(ValueError, match="`data` must be a 2D array."):
        PARRM(data=np.zeros((1, 1, 1)), sampling_freq=FS, artefact_freq=FA)
    with pytest.raises(TypeError, match="`sampling_freq` must be an int or a float."):
        PARRM(data=d, sampling_freq=str(FS), artefact_freq=FA)
    with pytest.raises(TypeError, match="`artefact_freq` must be an int or a float."):
        PARRM(data=d, sampling_freq=FS, artefact_freq=str(FA))
    with pytest.raises(ValueError, match="`sampling_freq` must be > 0."):
        PARRM(data=d, sampling_freq=0, artefact_freq=FA)
    with pytest.raises(ValueError, match="`artefact_freq` must be > 0."):
        PARRM(data=d, sampling_freq=FS, artefact_freq=0)
    with pytest.raises(TypeError, match="`verbose` must be a bool."):
        PARRM(data=d, sampling_freq=FS, artefact_freq=FA, verbose=str(False))
    # validation order: data type is checked before the frequencies
    with pytest.raises(TypeError, match="`data` must be a NumPy array."):
        PARRM(data=None, sampling_freq="x", artefact_freq="y")
    p = PARRM(d, FS, FA, verbose=False)
    assert p.data is d  # held by reference (parrm.py:124)
    assert (p._n_chans, p._n_samples) == (1, 100)


def test_find_period_input_validation():
    """All of these fail before any device work (parrm.py:213-270)."""
    d = _data()
    p = PARRM(d, FS, FA, verbose=False)
    with pytest.raises(TypeError, match="`search_samples` must be a NumPy array or None."):
        p.find_period(search_samples=0)
    with pytest.raises(ValueError, match="`search_samples` must be a 1D array."):
        p.find_period(search_samples=np.zeros((1, 1)))
    with pytest.raises(ValueError, match="Entries of `search_samples` must lie in the range "):
        p.find_period(search_samples=np.array([-1, 1]))
    with pytest.raises(ValueError, match="Entries of `search_samples` must lie in the range "):
        p.find_period(search_samples=np.array([0, d.shape[1]]))
    with pytest.raises(TypeError, match="`assumed_periods` must be an int, a float, a tuple, or None."):
        p.find_period(assumed_periods=[0])
    with pytest.raises(TypeError, match="If a tuple, entries of `assumed_periods` must be ints or "):
        p.find_period(assumed_periods=tuple(["test"]))
    with pytest.raises(TypeError, match="`outlier_boundary` must be an int or a float."):
        p.find_period(outlier_boundary=[0])
    with pytest.raises(ValueError, match="`outlier_boundary` must be > 0."):
        p.find_period(outlier_boundary=0)
    with pytest.raises(TypeError, match="`random_seed` must be an int or None."):
        p.find_period(random_seed=1.5)
    with pytest.raises(TypeError, match="`n_jobs` must be an int."):
        p.find_period(n_jobs=1.5)
    with pytest.raises(ValueError, match="`n_jobs` must be <= the number of available CPUs."):
        p.find_period(n_jobs=cpu_count() + 1)
    with pytest.raises(ValueError, match="If `n_jobs` is <= 0, it must be -1."):
        p.find_period(n_jobs=-2)


def test_premature_calls_state_machine():
    p = PARRM(_data(), FS, FA, verbose=False)
    with pytest.raises(ValueError, match="The period has not yet been estimated."):
        p.explore_filter_params()
    with pytest.raises(ValueError, match="The period has not yet been estimated."):
        p.create_filter()
    with pytest.raises(ValueError, match="The filter has not yet been created."):
        p.filter_data()
    with pytest.raises(AttributeError, match="No period has been computed yet."):
        p.period
    with pytest.raises(AttributeError, match="No filter has been computed yet."):
        p.filter
    with pytest.raises(AttributeError, match="No data has been filtered yet."):
        p.filtered_data
    with pytest.raises(AttributeError, match="Analysis settings have not been established yet."):
        p.settings
    p._period = np.float64(2.0239)
    with pytest.raises(ValueError, match="The filter has not yet been created."):
        p.filter_data()
    with pytest.raises(AttributeError, match="Analysis settings have not been established yet."):
        p.settings


def test_create_filter_validation_and_defaults():
    d = _data()
    p = _with_period(d, 2.023966953751087)
    with pytest.raises(TypeError, match="`filter_half_width` must be an int."):
        p.create_filter(filter_half_width=1.5)
    with pytest.raises(TypeError, match="`omit_n_samples` must be an int."):
        p.create_filter(omit_n_samples=1.5)
    with pytest.raises(TypeError, match="`filter_direction` must be a str."):
        p.create_filter(filter_direction=0)
    with pytest.raises(TypeError, match="`period_half_width` must be an int or a float."):
        p.create_filter(period_half_width=[0])
    with pytest.raises(ValueError, match="`filter_half_width` must lie in the range"):
        p.create_filter(filter_half_width=1, omit_n_samples=2)
    with pytest.raises(ValueError, match="`filter_half_width` must lie in the range"):
        p.create_filter(filter_half_width=((d.shape[1] - 1) // 2) + 1)
    with pytest.raises(ValueError, match="`omit_n_samples` must lie in the range"):
        p.create_filter(omit_n_samples=-1)
    with pytest.raises(ValueError, match="`omit_n_samples` must lie in the range"):
        p.create_filter(omit_n_samples=(d.shape[1] - 1) // 2)
    with pytest.raises(ValueError, match="`period_half_width` must be lie in the range "):
        p.create_filter(period_half_width=0)
    with pytest.raises(ValueError, match="`period_half_width` must be lie in the range "):
        p.create_filter(period_half_width=p.period + 1)
    with pytest.raises(ValueError, match="`filter_direction` must be one of "):
        p.create_filter(filter_direction="not_a_direction")
    with pytest.raises(RuntimeError, match="A suitable filter cannot be created with the specified "):
        p.create_filter(omit_n_samples=48)
    # order: omit_n_samples is checked before filter_half_width's type
    with pytest.raises(TypeError, match="`omit_n_samples` must be an int."):
        p.create_filter(filter_half_width=1.5, omit_n_samples=1.5)
    p.create_filter()
    assert p._filter_half_width == 49 and p._period_half_width == p.period / 50
    assert p.filter.shape == (99,) and p.filter[49] == 1
    assert p.filter is p._filter  # the internal array itself (parrm.py:901-905)
    s = p.settings
    assert s["data"] == {"sampling_freq": FS, "artefact_freq": FA}
    assert set(s["period"]) == {"search_samples", "assumed_periods", "outlier_boundary", "random_seed"}
    assert s["filter"] == {"filter_half_width": 49, "omit_n_samples": 0, "filter_direction": "both",
                           "period_half_width": p.period / 50}
    assert repr(p) == "PARRM object | Data: (1 channels x 100 times) | Period: 2.0240"


def test_create_filter_matches_reference_filters(golden):
    g = golden("filters.npz")
    n_samples = int(g["n_samples"])
    for tag in "abcdefgh":
        per, hw, omit, d, phw = g[f"{tag}_params"]
        p = _with_period(np.zeros((1, n_samples)), per, 22000, 130)
        explicit = tag in "afgh"
        p.create_filter(int(hw) if explicit else None, int(omit), DIRS[int(d)],
                        float(phw) if explicit else None)
        assert p._filter_half_width == int(hw)
        np.testing.assert_array_equal(p.filter, g[f"{tag}_filter"])


def test_filter_data_input_validation():
    p = _with_period(_data(), 2.023966953751087)
    p.create_filter()
    with pytest.raises(TypeError, match="`data` must be a NumPy array."):
        p.filter_data(data=_data().tolist())
    with pytest.raises(ValueError, match="`data` must be a 2D array."):
        p.filter_data(data=np.zeros(100))


def test_find_period_resets_downstream_state():
    p = _with_period(_data(), 2.023966953751087)
    p.create_filter()
    p._filtered_data = np.zeros((1, 100))
    p._reset_result_attrs()
    for name in ("_period", "_filter", "_filtered_data", "_filter_half_width", "_search_samples"):
        assert getattr(p, name) is None


def test_stage_indices_and_grid_match_reference(golden):
    """Host-side index draws / grids feed the device kernels; they must be the reference's."""
    g = golden("fit_errors_ecog.npz")
    ecog = np.load(get_example_data_paths("ecog_lfp_data"))
    p = PARRM(ecog, 1000, 130, verbose=False)
    p._check_sort_find_stim_period_inputs(None, None, 3.0, 44, 1)
    rng = np.random.default_rng(44)
    np.testing.assert_array_equal(p._get_centre_indices(5000, 0.0, rng), g["idx1"])
    p._get_centre_indices(10000, 0.0, rng)
    np.testing.assert_array_equal(p._get_centre_indices(25000, 0.95, rng), g["idx3"])
    from oracle import parrm_oracle as orc

    for run in (1, 2, 3):
        np.testing.assert_array_equal(
            p._get_possible_periods((7.69, 7.75), run), orc.possible_periods((7.69, 7.75), run)
        )
    assert p._n_jobs == 1
    p._check_sort_find_stim_period_inputs(None, 7.7, 3.0, None, -1)
    assert p._n_jobs == cpu_count() and p._assumed_periods == (7.7,)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_device_path_fails_loudly_without_gpu():
    """No CPU fallback: the product path raises instead of computing on the host."""
    p = PARRM(_data(), FS, FA, verbose=False)
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        p.find_period()
    p._period = np.float64(2.0239)
    p.create_filter()  # host-only: works
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        p.filter_data()
    with pytest.raises(_hip.HipLibraryError):
        _hip.FilterPlan(p.filter)


def test_product_never_imports_the_oracle():
    import pathlib
    import re

    root = pathlib.Path(__file__).resolve().parent.parent / "pyparrm_amd"
    for f in root.rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", f.read_text(), re.M), f


def test_explorer_argument_contract():
    """reference tests/test_parrm.py:158-180, :239-269 -- errors come before the GUI is declined."""
    d = _data()
    p = _with_period(d, 2.023966953751087)
    for kwargs, exc, msg in [
        (dict(time_range=0), TypeError, "`time_range` must be a list of ints or floats."),
        (dict(time_range=[0, "end"]), TypeError, "`time_range` must be a list of ints or floats."),
        (dict(time_res="all"), TypeError, "`time_res` must be an int or a float."),
        (dict(freq_range=0), TypeError, "`freq_range` must be a list of ints or floats."),
        (dict(freq_range=[0, "Nyquist"]), TypeError, "`freq_range` must be a list of ints or floats."),
        (dict(freq_res=[0]), TypeError, "`freq_res` must be an int or a float."),
        (dict(n_jobs=1.5), TypeError, "`n_jobs` must be an int."),
        (dict(time_range=[0, 1, 2]), ValueError, "`time_range` must have a length of 2."),
        (dict(time_range=[-1, 1]), ValueError, "`time_range` must lie in the range "),
        (dict(time_range=[0, d.shape[1] / FS + 1]), ValueError, "`time_range` must lie in the range "),
        (dict(time_range=[1, 0]), ValueError, "`time_range"),
        (dict(time_res=0), ValueError, "`time_res` must lie in the range "),
        (dict(time_res=d.shape[1] / FS), ValueError, "`time_res` must lie in the range "),
        (dict(freq_range=[0, 1, 2]), ValueError, "`freq_range` must have a length of 2."),
        (dict(freq_range=[-1, 1]), ValueError, "`freq_range` must lie in the range "),
        (dict(freq_range=[0, FS / 2 + 1]), ValueError, "`freq_range` must lie in the range "),
        (dict(freq_range=[1, 0]), ValueError, "`freq_range"),
        (dict(freq_res=0), ValueError, "`freq_res` must lie in the range "),
        (dict(freq_res=FS / 2 + 1), ValueError, "`freq_res` must lie in the range "),
        (dict(n_jobs=cpu_count() + 1), ValueError, "`n_jobs` must be <= the number of available CPUs."),
        (dict(n_jobs=-2), ValueError, "If `n_jobs` is <= 0, it must be -1."),
    ]:
        with pytest.raises(exc, match=msg):
            p.explore_filter_params(**kwargs)
    with pytest.raises(NotImplementedError):
        p.explore_filter_params()


def test_deepcopy_and_pickle_drop_device_state(golden):
    """The reference object is a plain Python object: its explorer starts with ``deepcopy(parrm)``
    (_utils/_plotting.py:115).  Host state must be copied, device state dropped (never pickled)."""
    import copy
    import pickle

    x = _data((2, 400))
    p = _with_period(x, 2.0)
    p.create_filter()
    sentinel = object()
    p._d_data, p._d_data_src, p._d_scale = sentinel, x, sentinel  # stand-ins for device state
    p._plans = {0: sentinel}
    p._last_plan = sentinel
    q = copy.deepcopy(p)
    assert q._d_data is None and q._d_scale is None and q._plans == {} and q._plan is None
    assert q._data is not p._data and np.array_equal(q._data, p._data)
    assert q.filter is not p.filter and np.array_equal(q.filter, p.filter)
    assert q.period == p.period and q.settings["filter"] == p.settings["filter"]
    # the explorer's next moves (_plotting.py:136-141, :186) work on the copy and leave the original alone
    q._data = q._data[:, 50:350]
    q._n_samples = q._data.shape[1]
    q._check_sort_create_filter_inputs(None, 0, "both", None)
    q._generate_filter()
    assert p._n_samples == 400 and p._data.shape == (2, 400)
    # pickling: same drop list
    p._plans, p._last_plan, p._d_data, p._d_scale = {}, None, None, None
    r = pickle.loads(pickle.dumps(p))
    assert np.array_equal(r.filter, p.filter) and r._plans == {} and r.period == p.period


def test_cache_on_device_flag_is_opt_in():
    p = PARRM(_data(), FS, FA, verbose=False)
    assert p.cache_on_device is False
    p.cache_on_device = True
    assert PARRM.cache_on_device is False


def test_online_filter_reach_and_latency_rules(golden):
    """Host-side rules of the online mode: a tap at offset w reads sample n - w, so the reference's
    ``"future"`` filters (w > 0, parrm.py:819-820) reach only EARLIER samples -- zero latency -- and
    ``"past"`` filters (w <= 0, :817-818) only later ones."""
    from pyparrm_amd.streaming import _tap_reach

    g = golden("filters.npz")
    for tag in "abcdefgh":
        filt = g[f"{tag}_filter"]
        hw, back, ahead = _tap_reach(filt)
        direction = DIRS[int(g[f"{tag}_params"][3])]
        assert hw == (filt.shape[0] - 1) // 2 and 0 <= back <= hw and 0 <= ahead <= hw
        if direction == "future":
            assert ahead == 0 and back > 0
        elif direction == "past":
            assert back == 0 and ahead > 0
        else:
            assert back > 0 and ahead > 0
    with pytest.raises(ValueError, match="no taps"):
        _tap_reach(np.array([0.0, 1.0, 0.0]))
    with pytest.raises(ValueError, match="odd length"):
        _tap_reach(np.zeros(4))
    # the stream object itself needs the device
    if not _has_gpu():
        p = _with_period(_data((1, 400)), 2.0)
        p.create_filter()
        with pytest.raises(_hip.HipLibraryError):
            p.online()
    q = PARRM(_data(), FS, FA, verbose=False)
    with pytest.raises(ValueError, match="The filter has not yet been created"):
        q.online()


def test_create_filter_random_parameters_match_the_oracle():
    """300 random (period, recording length, half-width, omitted samples, direction, period half-width) draws: the
    façade's ``create_filter`` (host NumPy, parrm.py:739-833) and the oracle's restatement -- written separately, the
    oracle pinned bit for bit by the reference's fixtures -- give the same half-width and the same taps, or raise the
    same RuntimeError."""
    from oracle import parrm_oracle as orc

    rng = np.random.default_rng(2024)
    checked = raised = 0
    for _ in range(300):
        period = float(rng.choice([1.3311, 2.024, 7.7424, 13.0, 64.3, 169.2358, 333.3])) * (1 + rng.uniform(-1e-3, 1e-3))
        n_samples = int(rng.choice([40, 400, 6001, 60000]))
        limit = (n_samples - 1) // 2
        omit = int(rng.integers(0, max(1, min(limit - 1, 40))))
        explicit_hw = rng.random() < 0.6
        hw = int(rng.integers(omit + 1, limit + 1)) if explicit_hw and limit > omit + 1 else None
        direction = str(rng.choice(["both", "past", "future"]))
        phw = float(rng.uniform(1e-3, 1.0) * period) if rng.random() < 0.6 else None
        p = _with_period(np.zeros((1, n_samples)), period, 22000, 130)
        phw_eff = period / 50 if phw is None else phw
        want_hw = hw if hw is not None else orc.default_filter_half_width(n_samples, period, omit, phw_eff)
        try:
            want = orc.generate_filter(period, want_hw, omit, direction, phw_eff)
        except RuntimeError:
            with pytest.raises(RuntimeError, match="A suitable filter cannot be created"):
                p.create_filter(hw, omit, direction, phw)
            raised += 1
            continue
        p.create_filter(hw, omit, direction, phw)
        assert p._filter_half_width == want_hw
        np.testing.assert_array_equal(p.filter, want)
        checked += 1
    assert checked > 200 and raised > 0


def test_ranking_by_selection_has_the_head_of_the_full_sort():
    """``_rank_candidates`` ranks a large all-finite grid by selection (the 7 smallest errors) instead of a full
    ``argsort`` (parrm.py:456-465): what the stage uses of the ranking -- the best <= 5 candidates in order, and the
    arg-min over [their refined errors, every other error] (:499-522) -- must be what the reference's full sort gives,
    on grids with exact ties (the full sort runs then), NaN / inf entries (ditto) and plain ones."""
    from pyparrm_amd import parrm as facade

    def reference_ranking(periods, errors):
        order = errors.argsort()
        e = errors[order]
        return periods[order[np.isfinite(e)]], e

    rng = np.random.default_rng(0)
    selected = 0
    for case in range(600):
        n = int(rng.integers(65, 12000))
        e = rng.standard_normal(n)
        if case % 5 == 1:
            e[rng.integers(0, n, 20)] = e[rng.integers(0, n, 20)]
        if case % 7 == 2:
            e[rng.integers(0, n)] = np.nan
        if case % 11 == 3:
            e[rng.integers(0, n)] = np.inf
        if case % 13 == 4:
            e = np.round(e, 2)
        per = rng.uniform(100, 200, n)
        got = facade._rank_candidates(per.copy(), e.copy())
        want = reference_ranking(per.copy(), e.copy())
        selected += got[0].shape[0] < want[0].shape[0]
        k = min(6, got[0].shape[0])
        assert np.array_equal(got[0][:k], want[0][:k]) and np.array_equal(got[1][:k], want[1][:k], equal_nan=True)
        refined = want[1][:5] + 0.1 * rng.standard_normal(5)
        a, b = [v.copy() for v in got], [v.copy() for v in want]
        a[1][:5] = refined
        b[1][:5] = refined
        if not np.isnan(b[1]).any():
            assert a[0][a[1].argmin()] == b[0][b[1].argmin()]
    assert selected > 300  # (the selection path is what ran in most cases)
