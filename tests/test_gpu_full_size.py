"""BASELINE configs[3] and configs[4] -- the two multi-GPU configurations -- with their arithmetic run at FULL SIZE on
the one device a test box has (``pytest -m gpu`` on an MI355X).

Why this is meaningful on one GPU: the reference's candidates are an ordered, independent map (parrm.py:445-454) and
its channels are filtered independently (parrm.py:861-866), so an N-rank run computes exactly what N thread-ranks on
one device compute -- the same kernels on the same blocks with the same launch geometry; only WHERE a block lives
differs.  The 1-rank search these tests compare with is the one ``test_config3_find_period_stage_by_stage_vs_oracle``
(tests/test_gpu_parity_r2.py) ties to the oracle piece by piece (same recording, same arguments).

Tolerances: configs[3] -- every comparison is ``==`` (bit-identical to the 1-rank run).  configs[4] -- float32 out:
max |y - oracle| <= 2e-6 * max |oracle| on the checked windows (north_star bar 1e-6 relative on float64; a float32
result carries one rounding of 6e-8 relative to ITS magnitude, inputs up to ~10 against outputs of order 1);
linearity within a few float32 roundings of the combined magnitudes; constant channels |y| <= 4e-15 |x|; repeat pass ``==``.
"""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import parrm_oracle as orc  # noqa: E402
from pyparrm_amd import PARRM, _hip  # noqa: E402
from pyparrm_amd import sharding as sh  # noqa: E402
from pyparrm_amd.synth import synth_recording_device  # noqa: E402

FS, FA = 22000.0, 130.0


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _hip.require_gpu()
    torch.cuda.set_device(0)


def _assumed_1e4():
    base = FS / FA
    return tuple(base * (1 + 0.02 * k) for k in range(-13, 13))


def _thread_ranks(x, world, assumed, seed):
    """``world`` thread-ranks on device 0, rank r holding channel block r of ``x`` (views, no copies): the whole
    path -- find_period, create_filter, filter_data -- through ``ShardedPARRM`` over a ``ThreadExchange``."""
    results, errors = [None] * world, []

    def rank_main(ex):
        try:
            torch.cuda.set_device(0)
            p = sh.ShardedPARRM(sh.shard_recording(x, ex.rank, world), FS, FA, ex, verbose=False)
            p.find_period(assumed_periods=assumed, random_seed=seed)
            p.create_filter()
            y = p.filter_data()
            results[ex.rank] = (p.period, [t["errors"] for t in p._trace[:3]], [len(t["grid"]) for t in p._trace[:3]], y)
        except BaseException as exc:  # noqa: BLE001 -- a failing rank would leave the others at the barrier
            errors.append(exc)
            ex._barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(ex,), name=f"rank{ex.rank}") for ex in sh.ThreadExchange.group(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


def test_config4_channel_sharded_256ch_10M_equals_one_rank_bitwise():
    """BASELINE configs[3]: 256 ch x 10 M float64 channel-sharded over 8, 4 and 2 ranks, the bench's search
    (26 assumed periods -> 10 044 + 387 + 381 candidates; stage-3 slices of ~48 / ~95 / ~190 candidates per rank)
    and the default filter.  Every rank's stage error vectors, its period and the concatenation of the ranks'
    ``filter_data`` blocks must equal the 1-rank ``PARRM`` on the whole recording bit for bit.  The single-process
    facade ``MultiDevicePARRM`` (one host thread per device, ``DeviceExchange``) is run the same way with the device
    repeated 8 times."""
    n_chans, n_samples = 256, 10_000_000
    x = synth_recording_device(n_chans, n_samples, FS, FA, seed=0)
    whole = PARRM(x, FS, FA, verbose=False)
    whole.find_period(assumed_periods=_assumed_1e4(), random_seed=44)
    assert [len(t["grid"]) for t in whole._trace[:3]] == [10044, 387, 381]
    assert abs(whole.period - FS / FA * (1 + 3e-5)) <= 2e-6 * whole.period
    whole.create_filter()
    y_whole = whole.filter_data()
    assert y_whole.is_cuda and y_whole.shape == x.shape
    want_errors = [t["errors"] for t in whole._trace[:3]]

    for world in (8, 4, 2):
        results = _thread_ranks(x, world, _assumed_1e4(), 44)
        for rank, (period, stage_errors, grid_sizes, y) in enumerate(results):
            assert period == whole.period, (world, rank, repr(period), repr(whole.period))
            assert grid_sizes == [10044, 387, 381]
            for stage, (got, want) in enumerate(zip(stage_errors, want_errors)):
                assert np.array_equal(got, want), (world, rank, stage)
            lo, hi = sh.channel_shard(n_chans, rank, world)
            assert y.shape == (hi - lo, n_samples)
            assert torch.equal(y, y_whole[lo:hi]), (world, rank)
        del results

    # the single-process facade over resident blocks: 8 "devices" (device 0 eight times), results stay on the device
    blocks = [sh.shard_recording(x, r, 8) for r in range(8)]
    multi = sh.MultiDevicePARRM.from_blocks(blocks, FS, FA, verbose=False)
    multi.find_period(assumed_periods=_assumed_1e4(), random_seed=44)
    assert multi.period == whole.period
    multi.create_filter()
    assert np.array_equal(multi.filter, whole.filter)
    ys = multi.filter_data()
    for r, y in enumerate(ys):
        lo, hi = sh.channel_shard(n_chans, r, 8)
        assert torch.equal(y, y_whole[lo:hi]), r


def _device_chunks(n_samples, chunk=12_500_000):
    return [(lo, min(lo + chunk, n_samples)) for lo in range(0, n_samples, chunk)]


def test_config5_one_gpu_share_128ch_50M_float32_streamed_from_pinned_host():
    """BASELINE configs[4], one GPU's share at its real size: 128 ch x 50 M float32 in page-locked host memory
    (25.6 GB in, 25.6 GB out), streamed through the device in the default 256 MiB time chunks with a half-width
    halo (``filter_host_sharded`` -> ``parrm_filter_host``), float32 out.

    * oracle windows on a sample of channels: both recording edges, the chunk seams (a chunk is
      2^19 samples of the 128 channels), random interior places;
    * constant channels are annihilated to the rounding of 1 / #taps (every output has at least one valid tap);
    * linearity: channels 0-63 hold a, channels 64-127 hold b; a second pass over [2.5 a - 0.75 b ; b] must give
      2.5 filter(a) - 0.75 filter(b) on the first half ...
    * ... and, on the unchanged second half, the first pass's bits (determinism of the streamed path)."""
    n_chans, n_samples = 128, 50_000_000
    period = FS / FA * (1 + 3e-5)
    hw = orc.default_filter_half_width(n_samples, period, 0, period / 50)
    assert hw == 2372
    filt = orc.generate_filter(period, hw, 0, "both", None)
    half = n_chans // 2
    const_rows = {5: 3.0, 77: -2.0}  # small integers: sums of up to 196 copies and their quotient are exact

    x_t = torch.empty((n_chans, n_samples), dtype=torch.float32, pin_memory=True)
    y1_t = torch.empty((n_chans, n_samples), dtype=torch.float32, pin_memory=True)
    y2_t = torch.empty((n_chans, n_samples), dtype=torch.float32, pin_memory=True)
    x, y1, y2 = x_t.numpy(), y1_t.numpy(), y2_t.numpy()
    # the recording: Gaussian background + a pulse train of the artefact period, generated on the device row by row
    gen = torch.Generator(device="cuda")
    n = torch.arange(n_samples, dtype=torch.float64, device="cuda")
    for c in range(n_chans):
        if c in const_rows:
            x_t[c].fill_(const_rows[c])
            continue
        gen.manual_seed(9000 + c)
        u = torch.remainder((n + 3.0 * c) / period, 1.0)
        art = torch.clamp(1.0 - torch.abs(u - 0.25) * 20.0, min=0.0) - 0.5 * torch.clamp(1.0 - torch.abs(u - 0.35) * 12.0, min=0.0)
        row = torch.randn(n_samples, dtype=torch.float32, device="cuda", generator=gen) + ((2.0 + c % 7) * art).to(torch.float32)
        x_t[c].copy_(row)
        del u, art, row
    del n
    torch.cuda.synchronize()

    out = sh.filter_host_sharded(filt, x, out=y1, devices=[0], out_dtype=np.float32)
    assert out is y1

    # ---- constant channels are annihilated: y = x - S * (1 / #taps) with S = #taps * x exact, so what is left is the
    # rounding of the reciprocal (a few 1e-16 of x), at the edges (fewer valid taps) as in the interior
    for c, value in const_rows.items():
        assert float(np.abs(y1[c]).max()) <= 4e-15 * abs(value), c

    # ---- oracle windows
    chunk = (256 << 20) // 4 // n_chans  # the default chunk: 256 MiB of float32 input = 524 288 samples of 128 channels
    assert chunk == 524_288
    rng = np.random.default_rng(11)
    span = 3000
    starts = [0, n_samples - span, chunk - span // 2, chunk - hw - 10, chunk + hw - span + 10, 2 * chunk - span // 2,
              95 * chunk - span // 2]  # (95 chunks and a remainder: the last seam)
    starts += [int(s) for s in rng.integers(hw, n_samples - span - hw, 5)]
    chans = [0, 1, 63, 64, 127] + [int(c) for c in rng.choice(n_chans, 6, replace=False)]
    chans = sorted(set(chans) - set(const_rows))
    scale = 0.0
    for s0 in starts:
        lo, hi = max(0, s0 - hw), min(n_samples, s0 + span + hw)
        ref = orc.filter_data_direct(x[chans, lo:hi].astype(np.float64), filt)
        got = y1[chans, lo:hi].astype(np.float64)
        keep = np.ones(hi - lo, dtype=bool)
        if lo > 0:
            keep[:hw] = False  # their taps reach samples in front of the window
        if hi < n_samples:
            keep[-hw:] = False
        scale = max(scale, float(np.abs(ref[:, keep]).max()))
        assert np.abs(got[:, keep] - ref[:, keep]).max() <= 2e-6 * np.abs(ref[:, keep]).max(), s0
    assert scale > 1.0

    # ---- second pass over [2.5 a - 0.75 b ; b]
    for c0 in range(0, half, 8):
        for lo, hi in _device_chunks(n_samples):
            a = x_t[c0:c0 + 8, lo:hi].cuda(non_blocking=True)
            b = x_t[half + c0:half + c0 + 8, lo:hi].cuda(non_blocking=True)
            x_t[c0:c0 + 8, lo:hi].copy_(a.mul_(2.5).add_(b, alpha=-0.75))
    torch.cuda.synchronize()
    sh.filter_host_sharded(filt, x, out=y2, devices=[0], out_dtype=np.float32)
    worst, worst_same = 0.0, True
    for c0 in range(0, half, 8):
        for lo, hi in _device_chunks(n_samples):
            ya = y1_t[c0:c0 + 8, lo:hi].cuda(non_blocking=True).double()
            yb = y1_t[half + c0:half + c0 + 8, lo:hi].cuda(non_blocking=True)
            yc = y2_t[c0:c0 + 8, lo:hi].cuda(non_blocking=True).double()
            yb2 = y2_t[half + c0:half + c0 + 8, lo:hi].cuda(non_blocking=True)
            worst_same = worst_same and bool(torch.equal(yb, yb2))
            want = ya.mul_(2.5).add_(yb.double(), alpha=-0.75)
            worst = max(worst, float((yc - want).abs().max()))
    assert worst_same, "a repeat pass over unchanged channels changed bits"
    # a, b ~ N(0, 1) + pulses up to 8, so |2.5 a - 0.75 b| <= ~45: the combined recording is rounded to float32 once
    # (6e-8 * 45 = 2.7e-6, passed through a filter of gain <= 2) and the three outputs once each
    assert worst <= 1.5e-5, worst
