"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures written from the unmodified reference.  Run with ``pytest -m gpu`` on an MI355X.

Tolerances (float64; BASELINE.json north_star: 1e-6 relative):
  filter_data   max|y - y_oracle| <= 1e-10 * max|y_oracle|  (asserted 4 orders tighter than 1e-6)
  fit errors    rtol 1e-9
  period        |T - T_ref| <= 1e-9 * T_ref and identical filter taps
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import parrm_oracle as orc  # noqa: E402
from pyparrm_amd import PARRM, _hip, get_example_data_paths  # noqa: E402
from pyparrm_amd.synth import synth_recording  # noqa: E402

FILTER_RTOL = 1e-10


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _hip.require_gpu()  # fails loudly (no skip) when the extension or the GPU is missing
    torch.cuda.set_device(0)


def _variants(plan):
    out = [_hip.KERNEL_GATHER]
    if plan.info.stride > 0:
        out.append(_hip.KERNEL_STRIDE)
    if plan.info.phase_groups > 0:
        out.append(_hip.KERNEL_PHASE)
    return out


def _run_filter(plan, x, kernel, out_dtype=None):
    plan.set_kernel(kernel)
    try:
        return plan.apply(torch.from_numpy(x).cuda(), out_dtype=out_dtype).cpu().numpy()
    finally:
        plan.set_kernel(_hip.KERNEL_AUTO)


def _assert_filter_close(y, ref, rtol=FILTER_RTOL, mask=None):
    scale = max(float(np.abs(ref).max()), 1e-300)
    diff = np.abs(y - ref)
    if mask is not None:
        diff = diff[:, mask]
    assert diff.size == 0 or diff.max() <= rtol * scale, f"max diff {diff.max():.3e} vs scale {scale:.3e}"


# ---------------------------------------------------------------------------- filter_data
def test_filter_matlab_known_answer(golden):
    """examples/plot_use_parrm.py:77-80,135-141,235-240 -- the reference's own numerical pin."""
    x = np.load(get_example_data_paths("example_data"))
    matlab = np.load(get_example_data_paths("matlab_filtered"))
    filt = golden("filter_data.npz")["example_filter"]
    plan = _hip.FilterPlan(filt)
    for k in _variants(plan):
        y = _run_filter(plan, x, k)
        assert np.allclose(y, matlab)  # the reference's check
        _assert_filter_close(y, matlab)
        _assert_filter_close(y, golden("filter_data.npz")["example_filtered"])


@pytest.mark.parametrize("tag", ["both", "past", "future"])
def test_filter_golden_synth(golden, tag):
    g = golden("filter_data.npz")
    x, filt, ref = g["synth_x"], g[f"synth_{tag}_filter"], g[f"synth_{tag}_y"]
    plan = _hip.FilterPlan(filt)
    ok = orc.valid_tap_counts(filt, x.shape[1]) > 0
    direct = orc.filter_data_direct(x, filt)
    for k in _variants(plan):
        y = _run_filter(plan, x, k)
        _assert_filter_close(y, ref, mask=ok)  # vs the reference (FFT noise where no tap is valid)
        _assert_filter_close(y, direct)        # vs the closed form, everywhere
        assert np.all(y[:, ~ok] == 0)
        # data shorter than the filter (tests/test_parrm.py:58-60)
        xs, ys = g[f"short_{tag}_x"], g[f"short_{tag}_y"]
        oks = orc.valid_tap_counts(filt, xs.shape[1]) > 0
        y_s = _run_filter(plan, xs, k)
        _assert_filter_close(y_s, ys, mask=oks)
        assert np.all(y_s[:, ~oks] == 0)


def test_filter_f32_promotes_to_f64(golden):
    g = golden("filter_data.npz")
    filt = g["synth_both_filter"]
    x32 = g["synth_x"].astype(np.float32)
    plan = _hip.FilterPlan(filt)
    ref = g["synth_f32_y"]  # the reference runs the float32 FFT in single precision: ~1e-6 noise
    direct = orc.filter_data_direct(x32, filt)  # float64 accumulation of the float32 samples
    for k in _variants(plan):
        y = _run_filter(plan, x32, k)
        assert y.dtype == np.float64
        _assert_filter_close(y, ref, rtol=1e-5)
        _assert_filter_close(y, direct)
        y32 = _run_filter(plan, x32, k, out_dtype=torch.float32)
        assert y32.dtype == np.float32
        # float32 -> float32 is a build option beyond the reference; its phase kernel sums the taps of a
        # row in float32 (packed adds): ~4e-7 of the sample scale, next to the float32 output's own 6e-8
        _assert_filter_close(y32.astype(np.float64), direct, rtol=2e-6)


FILTER_CASES = [
    # (period, hw, omit, direction, phw)
    (169.23584615384616, 2372, 0, "both", None),
    (169.23584615384616, 2372, 7, "past", None),
    (7.742402205597892, 2477, 0, "both", None),
    (7.742402205597892, 300, 10, "future", 0.5),
    (1.3311148014466094, 2000, 20, "both", 0.01),
    (50.0, 120, 0, "both", 50.0),       # every offset is a tap (box filter)
    (2.023966953751087, 49, 0, "both", None),  # two taps only
    (33.3, 7000, 0, "both", None),      # long half-width (large LDS ring)
    (169.2, 9000, 0, "both", None),     # half-width beyond the LDS ring: gather kernel only
]


@pytest.mark.parametrize("case", FILTER_CASES)
@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (3, 50), (1, 169), (5, 1000), (2, 20011), (3, 70001)])
def test_filter_shapes_and_taps(case, shape):
    period, hw, omit, direction, phw = case
    filt = orc.generate_filter(period, hw, omit, direction, phw)
    rng = np.random.default_rng(hash((shape, hw)) % (2**32))
    x = rng.standard_normal(shape) * 3.0 + 10.0  # DC offset stresses cancellation
    ref = orc.filter_data_direct(x, filt)
    plan = _hip.FilterPlan(filt)
    for k in _variants(plan):
        y = _run_filter(plan, x, k)
        assert y.shape == x.shape
        _assert_filter_close(y, ref, rtol=1e-10)


@pytest.mark.parametrize("period", [1.3311148014466094, 1.9230769, 2.0, 7.6923, 7.742402205597892, 13.0, 47.9])
def test_short_periods_take_the_phase_kernel(period):
    # the wrap form of the phase-major kernel: taps at several residues of the stride, per-lane wrap
    # to the previous row (parity of every variant is in test_filter_shapes_and_taps / fuzz_filter.py)
    n = 300_000
    hw = orc.default_filter_half_width(n, period, 0, period / 50)
    taps = orc.generate_filter(period, hw, 0, "both", None)
    plan = _hip.FilterPlan(taps)
    info = plan.info
    assert info.phase_groups > 0 and info.phase_guard == 0 and info.kernel == _hip.KERNEL_PHASE
    x = synth_recording(2, n, 22000, 130, seed=21)
    y = _run_filter(plan, x, _hip.KERNEL_AUTO)
    _assert_filter_close(y, orc.filter_data_direct(x, taps))


def test_filter_strided_rows_and_auto_kernel():
    filt = orc.generate_filter(169.23584615384616, 2372, 0, "both", None)
    rng = np.random.default_rng(5)
    big = rng.standard_normal((6, 90000))
    d_big = torch.from_numpy(big).cuda()
    view = d_big[:, 1000:81001]  # row stride 90000, 80001 samples, 8-byte-aligned only
    plan = _hip.FilterPlan(filt)
    assert plan.info.kernel == _hip.KERNEL_PHASE
    y = plan.apply(view).cpu().numpy()
    _assert_filter_close(y, orc.filter_data_direct(big[:, 1000:81001], filt))
    out = torch.zeros((6, 100000), dtype=torch.float64, device="cuda")
    plan.apply(view, out=out[:, 5:80006])
    assert torch.equal(out[:, 5:80006].cpu(), torch.from_numpy(y))
    assert float(out[:, :5].abs().sum()) == 0 and float(out[:, 80006:].abs().sum()) == 0


def test_filter_kernels_agree_bitwise_free_and_shard_invariant():
    """Channel shards are independent: filtering rows separately is bit-identical."""
    filt = orc.generate_filter(169.23584615384616, 2372, 0, "both", None)
    x = synth_recording(8, 150000, 22000, 130, seed=2)
    plan = _hip.FilterPlan(filt)
    d = torch.from_numpy(x).cuda()
    whole = plan.apply(d)
    parts = torch.cat([plan.apply(d[lo:lo + 2]) for lo in range(0, 8, 2)])
    assert torch.equal(whole, parts)
    again = plan.apply(d)
    assert torch.equal(whole, again)  # deterministic run to run


@pytest.mark.parametrize("shape", [None, "2,4", "2,3", "3,2"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_filter_interior_loop_is_bitwise_the_general_loop(monkeypatch, shape, dtype):
    # PARRM_DEBUG_FLAGS=16 keeps every iteration on the general loop; the lean interior loop (buffer
    # descriptors, scalar row offsets) must produce the same bits, for every workgroup shape
    if shape is None:
        monkeypatch.delenv("PARRM_PHASE_SHAPE", raising=False)
    else:
        monkeypatch.setenv("PARRM_PHASE_SHAPE", shape)
    x = synth_recording(3, 400_003, 22000, 130, seed=11).astype(dtype)
    taps = orc.generate_filter(169.2359, 2372, 0, "both")
    plan = _hip.FilterPlan(taps)
    monkeypatch.delenv("PARRM_PHASE_SHAPE", raising=False)
    assert plan.info.phase_groups > 0
    monkeypatch.delenv("PARRM_DEBUG_FLAGS", raising=False)
    fast = _run_filter(plan, x, _hip.KERNEL_PHASE)
    monkeypatch.setenv("PARRM_DEBUG_FLAGS", "16")
    general = _run_filter(plan, x, _hip.KERNEL_PHASE)
    monkeypatch.delenv("PARRM_DEBUG_FLAGS")
    assert np.array_equal(fast, general)
    _assert_filter_close(fast, orc.filter_data_direct(x.astype(np.float64), taps),
                         rtol=FILTER_RTOL if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("period,hw", [(169.23584615384616, 2372), (7.742402205597892, 2477), (1.9230769, 1250)])
def test_filter_window_and_host_streaming(period, hw):
    filt = orc.generate_filter(period, hw, 3, "both", None)
    x = synth_recording(3, 60000, 22000, 130, seed=4)
    plan = _hip.FilterPlan(filt)
    whole = plan.apply(torch.from_numpy(x).cuda()).cpu().numpy()
    n_total = x.shape[1]
    pieces = []
    for o0 in range(0, n_total, 17000):
        olen = min(17000, n_total - o0)
        b0, b1 = max(o0 - hw, 0), min(o0 + olen + hw, n_total)
        xb = torch.from_numpy(np.ascontiguousarray(x[:, b0:b1])).cuda()
        pieces.append(plan.apply_window(xb, b0, o0, olen, n_total).cpu().numpy())
    chunked = np.concatenate(pieces, axis=1)
    _assert_filter_close(chunked, whole, rtol=1e-12)
    streamed = plan.apply_host(x, chunk_samples=13001)
    _assert_filter_close(streamed, whole, rtol=1e-12)
    streamed32 = plan.apply_host(x.astype(np.float32), out_dtype=np.float32, chunk_samples=20000)
    _assert_filter_close(streamed32.astype(np.float64), whole, rtol=1e-5)
    # page-locked buffers of the caller's (hipHostMalloc through torch's pinned allocator) are used in place: same bits.
    # parrm_host_pin / parrm_host_unpin are no-ops by default (hipHostRegister + hipHostUnregister of heap pages made
    # the NEXT pageable copy through those pages fault on this ROCm build -- include/parrm_hip.h).
    x_locked = torch.from_numpy(x).pin_memory().numpy()
    out = torch.empty(x.shape, dtype=torch.float64).pin_memory().numpy()
    _hip.pin_host(out)  # already page-locked: left alone
    for _ in range(2):
        got = plan.apply_host(x_locked, chunk_samples=13001, out=out)
        assert got is out and np.array_equal(out, streamed)
    _hip.unpin_host(out)
    plain = np.empty_like(x)
    _hip.pin_host(plain)  # not locked (no PARRM_HOST_LOCK): the call stages it
    assert np.array_equal(plan.apply_host(x, chunk_samples=13001, out=plain), streamed)
    _hip.unpin_host(plain)
    with pytest.raises(ValueError):
        plan.apply_host(x, out=np.empty((2, 5)))


def test_filter_nonfinite_outputs_are_zeroed():
    filt = orc.generate_filter(20.0, 100, 0, "both", None)
    x = np.random.default_rng(1).standard_normal((2, 3000))
    x[0, 1500] = np.inf
    plan = _hip.FilterPlan(filt)
    for k in _variants(plan):
        y = _run_filter(plan, x, k)
        assert np.all(np.isfinite(y))
        np.testing.assert_allclose(y[1], orc.filter_data_direct(x[1:], filt)[0], rtol=0, atol=1e-12)


@pytest.mark.parametrize("period,hw,direction", [(169.23584615384616, 2372, "both"), (7.742402205597892, 2477, "both"),
                                                 (169.23584615384616, 2372, "future")])
def test_filter_nonfinite_samples_only_zero_the_outputs_they_reach(period, hw, direction):
    """the recurrence kernels carry running sums, so a NaN/Inf sample used to wipe every later
    output of its residue class up to the end of the stretch.  Now every kernel variant, the window form,
    the host-streamed form and the online form give the oracle's answer: 0 exactly where a tap reaches a
    non-finite sample (parrm.py:869), the exact value everywhere else -- on the poisoned channel too."""
    filt = orc.generate_filter(period, hw, 0, direction, None)
    n = 400_000
    x = synth_recording(3, n, 22000, 130, seed=12)
    x[0, 123_456] = np.nan
    x[0, 300_000:300_003] = np.inf
    x[2, 5] = -np.inf
    x[2, n - 2] = np.nan
    ref = orc.filter_data_direct(x, filt)
    assert np.count_nonzero(ref[0] == 0) < 4 * (2 * hw + 1) + 8  # only the reach of the bad samples is zeroed
    plan = _hip.FilterPlan(filt)
    for k in _variants(plan):
        y = _run_filter(plan, x, k)
        assert np.all(np.isfinite(y))
        _assert_filter_close(y, ref)
        assert np.array_equal(y == 0, ref == 0), k
    streamed = plan.apply_host(x, chunk_samples=90_001)
    _assert_filter_close(streamed, ref)
    assert np.array_equal(streamed == 0, ref == 0)
    from pyparrm_amd.streaming import OnlineFilter

    stream = OnlineFilter(filt, 3)
    parts = [stream.push(x[:, lo:lo + 70_000]) for lo in range(0, n, 70_000)] + [stream.finish()]
    online = np.concatenate(parts, axis=1)
    _assert_filter_close(online, ref)
    assert np.array_equal(online == 0, ref == 0)


# ---------------------------------------------------------------------------- find_period pieces
def test_absdiff_mean_and_gather(golden):
    g = golden("standardise.npz")
    x = g["x"]
    d = torch.from_numpy(x).cuda()
    scale = _hip.absdiff_mean(d)
    np.testing.assert_allclose(scale.cpu().numpy(), np.abs(np.diff(x, axis=1)).mean(axis=1), rtol=1e-13)
    idx = np.arange(x.shape[1] - 1)
    y = _hip.gather_standardise(d, torch.from_numpy(idx).cuda(), scale, float(g["outlier_boundary"]))
    np.testing.assert_allclose(y.cpu().numpy().T, g["std"], rtol=1e-13, atol=1e-15)
    # larger, ragged length, strided rows
    rng = np.random.default_rng(8)
    big = rng.standard_normal((5, 200003)) * np.array([[1], [3], [0.1], [20], [7]])
    dv = torch.from_numpy(big).cuda()[:, 3:]
    scale = _hip.absdiff_mean(dv).cpu().numpy()
    np.testing.assert_allclose(scale, np.abs(np.diff(big[:, 3:], axis=1)).mean(axis=1), rtol=1e-13)
    b32 = big.astype(np.float32)
    s32 = _hip.absdiff_mean(torch.from_numpy(b32).cuda()).cpu().numpy()
    ref32 = np.abs(np.diff(b32, axis=1)).astype(np.float64).mean(axis=1)
    np.testing.assert_allclose(s32, ref32, rtol=1e-12)


def _ecog_stage(indices):
    ecog = np.load(get_example_data_paths("ecog_lfp_data"))
    d = torch.from_numpy(ecog).cuda()
    scale = _hip.absdiff_mean(d)
    d_idx = torch.from_numpy(np.ascontiguousarray(indices)).cuda()
    return _hip.gather_standardise(d, d_idx, scale, 3.0), d_idx


def test_fit_errors_golden(golden):
    # measured against this fixture (scripts/fit_error_accuracy.py): 5.5e-15 (K = 11), 7.5e-14 (K = 41), 9.2e-12 (K = 41,
    # lambda 0) with the default design matrix; 3.1e-16 / 1.7e-14 / 2.1e-12 with PARRM_FIT_EXACT_TRIG=1
    g = golden("fit_errors_ecog.npz")
    y, d_idx = _ecog_stage(g["idx1"])
    np.testing.assert_allclose(y.cpu().numpy().T[:, :64], g["std_cols1"], rtol=1e-13, atol=1e-15)
    e1 = _hip.fit_errors(y, d_idx, g["per1"], 5, 1.0)
    np.testing.assert_allclose(e1, g["err1"], rtol=1e-10)
    e10 = _hip.fit_errors(y, d_idx, np.array([7.7424]), 10, 1.0)
    np.testing.assert_allclose(e10[0], float(g["err_arr"]), rtol=1e-10)
    y3, d_idx3 = _ecog_stage(g["idx3"])
    e3 = _hip.fit_errors(y3, d_idx3, g["per3"], 20, 1.0)
    np.testing.assert_allclose(e3, g["err3"], rtol=1e-10)
    e30 = _hip.fit_errors(y3, d_idx3, g["per3"], 20, 0.0)
    np.testing.assert_allclose(e30, g["err3_l0"], rtol=1e-10)
    # one candidate at a time (the Nelder-Mead regime: sample-split + reduce) == batched
    single = np.array([_hip.fit_errors(y3, d_idx3, g["per3"][i:i + 1], 20, 1.0)[0] for i in range(3)])
    np.testing.assert_allclose(single, e3[:3], rtol=1e-12)


def test_fit_errors_many_channels_vs_oracle():
    x = synth_recording(300, 12000, 22000, 130, seed=3)  # > 256 channels: two column blocks
    std = orc.standardise_data(x, 3.0)
    idx = np.arange(3000, 8001)
    d = torch.from_numpy(x).cuda()
    scale = _hip.absdiff_mean(d)
    d_idx = torch.from_numpy(idx).cuda()
    y = _hip.gather_standardise(d, d_idx, scale, 3.0)
    periods = 169.2359 * (1 + np.linspace(-1e-3, 1e-3, 5))
    for bw, lam in ((5, 1.0), (10, 1.0), (20, 0.0)):
        ref = orc.grid_errors(periods, std, idx, bw, lam)
        out = _hip.fit_errors(y, d_idx, periods, bw, lam)
        np.testing.assert_allclose(out, ref, rtol=1e-9)


def test_fit_errors_matrix_core_and_vector_paths_agree(monkeypatch):
    # the MFMA Gram kernel (n_chans % 4 == 0) against the vector-ALU kernel and the oracle; 5 channels
    # can only take the vector kernel
    x = synth_recording(8, 12000, 22000, 130, seed=5)
    std = orc.standardise_data(x, 3.0)
    idx = np.arange(2000, 9003)
    d = torch.from_numpy(x).cuda()
    scale = _hip.absdiff_mean(d)
    d_idx = torch.from_numpy(idx).cuda()
    y = _hip.gather_standardise(d, d_idx, scale, 3.0)
    periods = 169.2359 * (1 + np.linspace(-2e-3, 2e-3, 9))
    for bw, lam in ((5, 1.0), (10, 1.0), (20, 1.0), (7, 0.5)):
        ref = orc.grid_errors(periods, std, idx, bw, lam)
        monkeypatch.delenv("PARRM_FIT_ACCUM", raising=False)
        mfma = _hip.fit_errors(y, d_idx, periods, bw, lam)
        monkeypatch.setenv("PARRM_FIT_ACCUM", "1")
        valu = _hip.fit_errors(y, d_idx, periods, bw, lam)
        monkeypatch.delenv("PARRM_FIT_ACCUM")
        np.testing.assert_allclose(mfma, ref, rtol=1e-9)
        np.testing.assert_allclose(valu, ref, rtol=1e-9)
        np.testing.assert_allclose(mfma, valu, rtol=1e-11)
    # 5 channels: a contiguous [n, 5] matrix (odd row stride) can only take the vector kernel; the
    # same columns inside rows padded to whole quads take the matrix cores (quad straddles n_chans)
    ref5 = orc.grid_errors(periods, std[:5], idx, 5, 1.0)
    y5 = y[:, :5].contiguous()
    np.testing.assert_allclose(_hip.fit_errors(y5, d_idx, periods, 5, 1.0), ref5, rtol=1e-9)
    np.testing.assert_allclose(_hip.fit_errors(y[:, :5], d_idx, periods, 5, 1.0), ref5, rtol=1e-9)
    for n_ch in (1, 2, 3, 6, 7):
        xs = x[:n_ch]
        ds = torch.from_numpy(np.ascontiguousarray(xs)).cuda()
        ys = _hip.gather_standardise(ds, d_idx, _hip.absdiff_mean(ds), 3.0)
        assert ys.shape == (idx.shape[0], n_ch) and ys.stride(0) % 4 == 0
        refs = orc.grid_errors(periods, orc.standardise_data(xs, 3.0), idx, 10, 1.0)
        np.testing.assert_allclose(_hip.fit_errors(ys, d_idx, periods, 10, 1.0), refs, rtol=1e-9)
    # a strided stage matrix (ldy > n_chans) on the matrix-core path
    y_wide = torch.zeros((y.shape[0], 12), dtype=torch.float64, device="cuda")
    y_wide[:, :8] = y
    ref8 = orc.grid_errors(periods, std, idx, 10, 1.0)
    np.testing.assert_allclose(_hip.fit_errors(y_wide[:, :8], d_idx, periods, 10, 1.0), ref8, rtol=1e-9)


# ---------------------------------------------------------------------------- find_period
def _same_taps(p_gpu, p_ref, n_samples, **kw):
    hw = orc.default_filter_half_width(n_samples, p_ref, 0, p_ref / 50)
    a = orc.generate_filter(p_gpu, hw, 0, "both", p_gpu / 50)
    b = orc.generate_filter(p_ref, hw, 0, "both", p_ref / 50)
    return np.array_equal(a != 0, b != 0)


def test_period_example_data(golden):
    ref = float(golden("periods.npz")["example_data"])
    x = np.load(get_example_data_paths("example_data"))
    p = PARRM(x, 200, 150, verbose=False)
    p.find_period()
    assert isinstance(p.period, np.float64)
    assert p.period == ref, (repr(p.period), repr(ref))  # bit-identical, not 1e-9: taps are a step function of T
    # the full reference flow reproduces the MATLAB output (plot_use_parrm.py:135-141,239)
    p.create_filter(filter_half_width=2000, omit_n_samples=20, filter_direction="both",
                    period_half_width=0.01)
    assert np.array_equal(p.filter, golden("filter_data.npz")["example_filter"])
    y = p.filter_data()
    assert np.allclose(y, np.load(get_example_data_paths("matlab_filtered")))


def test_period_ecog_seed44(golden):
    ref = float(golden("periods.npz")["ecog_lfp_data_seed44"])
    x = np.load(get_example_data_paths("ecog_lfp_data"))
    p = PARRM(x, 1000, 130, verbose=False)
    p.find_period(random_seed=44)
    assert p.period == ref, (repr(p.period), repr(ref))  # bit-identical, not 1e-9: taps are a step function of T
    assert _same_taps(p.period, ref, x.shape[1])


def test_period_synth_and_options(golden):
    g = golden("periods.npz")
    p = PARRM(g["synth_4x30000"], 22000, 130, verbose=False)
    p.find_period(random_seed=3)
    ref = float(g["synth_4x30000_seed3"])
    assert p.period == ref, (repr(p.period), repr(ref))  # bit-identical, not 1e-9: taps are a step function of T
    assert _same_taps(p.period, ref, 30000)
    p2 = PARRM(g["synth_3x6000"], 1000, 130, verbose=False)
    p2.find_period(search_samples=np.arange(0, 3000.0), assumed_periods=(7.6, 7.7), random_seed=1)
    ref2 = float(g["synth_3x6000_half_two_estimates"])
    assert p2.period == ref2, (repr(p2.period), repr(ref2))


def test_period_nonfinite_data_raises():
    x = np.random.default_rng(0).standard_normal((1, 3000))
    x[0, 100] = np.nan
    p = PARRM(x, 1000, 130, verbose=False)
    with pytest.raises(ValueError, match="The period cannot be estimated from the data."):
        p.find_period()


# ---------------------------------------------------------------------------- facade smoke (reference tests/test_parrm.py:17-68)
@pytest.mark.parametrize("n_chans", [1, 2])
@pytest.mark.parametrize("n_samples", [100, 300, 25000])
@pytest.mark.parametrize("half", [False, True])
def test_facade_reference_smoke(n_chans, n_samples, half):
    rng = np.random.default_rng(44)
    data = rng.standard_normal((n_chans, n_samples))
    p = PARRM(data=data, sampling_freq=20, artefact_freq=10, verbose=False)
    search = np.arange(0, n_samples * 0.5) if half else None
    p.find_period(search_samples=search, assumed_periods=20 / 10, random_seed=44, n_jobs=2)
    for direction in ["future", "past", "both"]:
        p.create_filter(filter_direction=direction)
    y = p.filter_data()
    assert y.shape == data.shape and isinstance(y, np.ndarray) and y.dtype == np.float64
    assert y is p.filtered_data
    _assert_filter_close(y, orc.filter_data_direct(data, p.filter))
    other = rng.standard_normal((1, 50))
    assert p.filter_data(other).shape == other.shape
    assert repr(p) == (
        f"PARRM object | Data: ({n_chans} channels x {n_samples} times) | Period: {p.period :.4f}"
    )


def test_explorer_protocol_on_private_surface():
    # what the reference's parameter explorer does to a PARRM object (_plotting.py:111-200, 360-361,
    # 568-584): silence it, rebind `_data` to a time slice, re-validate defaults, then per widget event
    # poke the four filter settings, regenerate the filter and re-filter
    x = synth_recording(2, 60000, 22000, 130, seed=9)
    p = PARRM(x, 22000, 130, verbose=True)
    p._period = np.float64(169.2359)
    assert p._period is not None
    p._verbose = False
    time_range = np.arange(5000, 45000)
    p._data = p._data[:, time_range]
    p._n_samples = p._data.shape[1]
    p._check_sort_create_filter_inputs(None, 0, "both", None)
    assert (p._period_half_width, p._filter_direction, p._omit_n_samples) == (p._period / 50, "both", 0)
    hw0 = p._filter_half_width
    for hw, phw, omit, direction in ((hw0, p._period / 50, 0, "both"), (1500, 2.0, 10, "past"),
                                      (900, 4.5, 0, "future"), (hw0, 1.0, 3, "both")):
        p._filter_half_width, p._period_half_width = hw, phw
        p._omit_n_samples, p._filter_direction = omit, direction
        p._generate_filter()
        out = p.filter_data()
        taps = orc.generate_filter(p._period, hw, omit, direction, phw)
        assert np.array_equal(p.filter, taps)
        assert out.shape == (p._n_chans, p._n_samples)
        _assert_filter_close(out, orc.filter_data_direct(x[:, time_range], taps))


def test_facade_device_tensor_roundtrip():
    x = synth_recording(4, 50000, 22000, 130, seed=6)
    d = torch.from_numpy(x).cuda()
    p = PARRM(d, 22000, 130, verbose=False)
    p.find_period(random_seed=2)
    p.create_filter()
    y = p.filter_data()
    assert isinstance(y, torch.Tensor) and y.is_cuda and y.dtype == torch.float64
    _assert_filter_close(y.cpu().numpy(), orc.filter_data_direct(x, p.filter))


# ---------------------------------------------------------------------------- BASELINE-size properties
def test_config2_properties_64ch_1M():
    """BASELINE config 2 (64 ch x 1 Msample f64): size-independent properties + sampled oracle."""
    n_chans, n_samples = 64, 1_000_000
    filt = orc.generate_filter(169.23584615384616, 2372, 0, "both", None)
    plan = _hip.FilterPlan(filt)
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn((n_chans, n_samples), dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn((n_chans, n_samples), dtype=torch.float64, device="cuda", generator=g)
    ya, yb = plan.apply(a), plan.apply(b)
    # linearity
    yc = plan.apply(2.5 * a - 0.75 * b)
    assert float((yc - (2.5 * ya - 0.75 * yb)).abs().max()) < 1e-11
    # a constant is its own phase-neighbourhood mean -> exactly removable
    ones = torch.full((2, n_samples), 7.25, dtype=torch.float64, device="cuda")
    assert float(plan.apply(ones).abs().max()) < 1e-12
    # a signal with an exactly periodic integer period that the taps hit is annihilated inside
    per = 13
    f13 = orc.generate_filter(float(per), 650, 0, "both", 0.25)
    ramp = (torch.arange(n_samples, device="cuda") % per).to(torch.float64)[None, :].repeat(2, 1)
    y13 = _hip.FilterPlan(f13).apply(ramp)
    assert float(y13.abs().max()) < 1e-12
    # sampled closed-form check of three channels
    rows = [0, 31, 63]
    ref = orc.filter_data_direct(a[rows].cpu().numpy(), filt)
    _assert_filter_close(ya[rows].cpu().numpy(), ref)
    # gather kernel == stride kernel to rounding on a slice
    plan.set_kernel(_hip.KERNEL_GATHER)
    yg = plan.apply(a[:2])
    plan.set_kernel(_hip.KERNEL_AUTO)
    assert float((yg - ya[:2]).abs().max()) < 1e-11


def test_config3_properties_256ch_10M():
    """BASELINE config 3 (256 ch x 10 Msample f64, the bench workload, 61 GB of device buffers):
    size-independent properties at full size + closed-form checks of whole rows and of windows at
    the stretch seams, the recording edges and random places."""
    n_chans, n_samples = 256, 10_000_000
    filt = orc.generate_filter(169.23584615384616, 2372, 0, "both", None)
    hw = 2372
    plan = _hip.FilterPlan(filt)
    assert plan.info.kernel == _hip.KERNEL_PHASE
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((n_chans, n_samples), dtype=torch.float64, device="cuda", generator=g)
    ya = plan.apply(a)
    # linearity, through in-place combinations (no fourth full-size buffer)
    b = torch.randn((n_chans, n_samples), dtype=torch.float64, device="cuda", generator=g)
    yb = plan.apply(b)
    b.mul_(-0.75).add_(a, alpha=2.5)          # b <- 2.5 a - 0.75 b
    yb.mul_(-0.75).add_(ya, alpha=2.5)        # expected filter(b)
    yc = plan.apply(b)
    assert float((yc - yb).abs().max()) < 1e-11
    del b, yb, yc
    # determinism: a second launch gives the same bits
    assert torch.equal(plan.apply(a), ya)
    # whole rows against the closed form
    rows = [0, 100, 255]
    ref = orc.filter_data_direct(a[rows].cpu().numpy(), filt)
    _assert_filter_close(ya[rows].cpu().numpy(), ref)
    # windows of every channel: edges, the first stretch seams, random interior places
    stretch = (262144 // 169 // 8 * 8) * 169  # rows per stretch are a multiple of NG*R = 8
    rng = np.random.default_rng(5)
    starts = [0, n_samples - 4000, stretch - 2000, 2 * stretch - 2000] + list(rng.integers(hw, n_samples - 4000 - hw, 6))
    chans = torch.from_numpy(rng.permutation(n_chans)[:16]).cuda()
    for s0 in starts:
        lo, hi = max(0, s0 - hw), min(n_samples, s0 + 4000 + hw)
        xs = a[chans][:, lo:hi].cpu().numpy()
        # closed form on the window; only positions whose taps all lie inside the window (or at a
        # true recording edge) are comparable
        ref = orc.filter_data_direct(xs, filt)
        got = ya[chans][:, lo:hi].cpu().numpy()
        keep = np.ones(hi - lo, dtype=bool)
        if lo > 0:
            keep[:hw] = False
        if hi < n_samples:
            keep[-hw:] = False
        _assert_filter_close(got, ref, mask=keep)


# ---------------------------------------------------------------------------- input robustness
def test_facade_input_kinds():
    """float32 / integer / non-contiguous recordings behave like the reference's dtype rules."""
    fs, fa = 22000, 130
    x = synth_recording(3, 40000, fs, fa, seed=9)
    ref_p = PARRM(x, fs, fa, verbose=False)
    ref_p.find_period(random_seed=5)
    ref_p.create_filter()
    y_ref = ref_p.filter_data()
    # float32 recording: period close (statistics in float64 of float32 diffs), output float64
    p32 = PARRM(x.astype(np.float32), fs, fa, verbose=False)
    p32.find_period(random_seed=5)
    assert abs(p32.period - ref_p.period) < 1e-6 * ref_p.period
    p32._period = ref_p.period
    p32.create_filter()
    y32 = p32.filter_data()
    assert y32.dtype == np.float64
    _assert_filter_close(y32, orc.filter_data_direct(x.astype(np.float32), ref_p.filter))
    # integer recording: filter_data promotes through float64 (reference: convolve of ints);
    # find_period raises TypeError like the reference's in-place divide (parrm.py:275)
    xi = np.round(x * 100).astype(np.int64)
    pi = PARRM(xi, fs, fa, verbose=False)
    with pytest.raises(TypeError):
        pi.find_period()
    pi._period = ref_p.period
    pi.create_filter()
    _assert_filter_close(pi.filter_data(), orc.filter_data_direct(xi.astype(np.float64), ref_p.filter))
    # Fortran-ordered / sliced view
    xf = np.asfortranarray(x)
    pf = PARRM(xf, fs, fa, verbose=False)
    pf._period = ref_p.period
    pf.create_filter()
    assert np.array_equal(pf.filter_data(), y_ref)
    # other data through the same object, then the cached recording again
    other = synth_recording(2, 9000, fs, fa, seed=10)
    _assert_filter_close(ref_p.filter_data(other), orc.filter_data_direct(other, ref_p.filter))
    assert np.array_equal(ref_p.filter_data(), y_ref)


def test_find_period_is_deterministic_and_refindable():
    x = synth_recording(2, 30000, 22000, 130, seed=12)
    p = PARRM(x, 22000, 130, verbose=False)
    p.find_period(random_seed=3)
    first = p.period
    p.create_filter()
    p.filter_data()
    p.find_period(random_seed=3)  # resets downstream state (parrm.py:196-211)
    assert p.period == first
    with pytest.raises(AttributeError):
        p.filter
    with pytest.raises(AttributeError):
        p.filtered_data
