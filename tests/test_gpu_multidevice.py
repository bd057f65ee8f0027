"""Round-3 GPU tests of the multi-GPU paths that need no launcher (``pytest -m gpu``): ``sharding.MultiDevicePARRM``
(one process, one host thread per device, peer copies -- SURVEY.md section 7 step 6) and ``filter_host_sharded``
(BASELINE configs[4] as a package call).  On a one-GPU box the ranks share the device (``devices=[0, 0, 0]``);
where more devices are visible the same tests spread over them."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import parrm_oracle as orc  # noqa: E402
from pyparrm_amd import PARRM, _hip, sharding  # noqa: E402
from pyparrm_amd.synth import synth_recording, synth_recording_exact  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _hip.require_gpu()
    torch.cuda.set_device(0)


def _device_lists():
    n = torch.cuda.device_count()
    lists = [[0, 0, 0], [0, 0]]
    if n > 1:
        lists.append(list(range(n)))
    return lists


def test_multi_device_facade_is_bit_identical_to_one_device():
    """configs[3] without a launcher: NumPy in, NumPy out; period and filtered recording equal the single-device
    ``PARRM`` bit for bit (candidate grids planned as whole grids, time axis cut as for the whole recording)."""
    fs, fa = 22000.0, 130.0
    x = synth_recording_exact(7, 90_000, fs / fa * (1 + 2e-5), seed=21)
    one = PARRM(x, fs, fa, verbose=False)
    one.find_period(random_seed=3)
    one.create_filter()
    ref = one.filter_data()
    for devices in _device_lists():
        p = sharding.MultiDevicePARRM(x, fs, fa, devices=devices, verbose=False)
        p.find_period(random_seed=3)
        assert p.period == one.period, devices
        p.create_filter()
        assert np.array_equal(p.filter, one.filter)
        y = p.filter_data()
        assert isinstance(y, np.ndarray) and np.array_equal(y, ref), devices
        other = np.ascontiguousarray(x[:5, :40_000])
        assert np.array_equal(p.filter_data(other), one.filter_data(other)), devices
    with pytest.raises(TypeError):
        sharding.MultiDevicePARRM([1, 2, 3], fs, fa)  # the reference's own argument checks come first


def test_multi_device_resident_blocks_stay_on_their_devices():
    """The benchmark's form: channel blocks already in HBM, results returned per device."""
    fs, fa = 22000.0, 130.0
    x = synth_recording_exact(6, 70_000, fs / fa * (1 + 2e-5), seed=22)
    one = PARRM(x, fs, fa, verbose=False)
    one.find_period(random_seed=4)
    one.create_filter()
    ref = one.filter_data()
    n = max(2, torch.cuda.device_count())
    devs = [r % torch.cuda.device_count() for r in range(n)]
    blocks = [torch.from_numpy(np.ascontiguousarray(sharding.shard_recording(x, r, n))).to(f"cuda:{devs[r]}") for r in range(n)]
    p = sharding.MultiDevicePARRM.from_blocks(blocks, fs, fa, verbose=False)
    p.find_period(random_seed=4)
    assert p.period == one.period
    p.create_filter()
    outs = p.filter_data()
    assert all(o.is_cuda and o.device == b.device for o, b in zip(outs, blocks))
    assert np.array_equal(np.concatenate([o.cpu().numpy() for o in outs]), ref)


def test_two_searches_on_one_device_from_two_threads_do_not_share_staging():
    """the page-locked index staging vector is per device, not per search -- two threads running
    `find_period` with different seeds on one device must each get their own indices."""
    import threading

    fs, fa = 22000.0, 130.0
    recs = [synth_recording_exact(2, 80_000, fs / fa * (1 + 2e-5), seed=30 + i) for i in range(2)]
    serial = []
    for i, x in enumerate(recs):
        p = PARRM(x, fs, fa, verbose=False)
        p.find_period(random_seed=10 + i)
        serial.append(p.period)
    got = [None, None]

    def work(i):
        torch.cuda.set_device(0)
        for _ in range(3):
            p = PARRM(recs[i], fs, fa, verbose=False)
            p.find_period(random_seed=10 + i)
            got[i] = p.period if got[i] in (None, p.period) else float("nan")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == serial


@pytest.mark.parametrize("dtype,out_dtype,tol", [(np.float32, np.float32, 1e-5), (np.float64, np.float64, 1e-10)])
def test_filter_host_sharded_matches_the_oracle(dtype, out_dtype, tol):
    """configs[4] in miniature through the package call: host recording, channel blocks on thread-ranks, each
    block streamed in time chunks with a half-width halo; the oracle checks every channel."""
    period = 22000 / 130 * (1 + 3e-5)
    n = 700_000
    hw = orc.default_filter_half_width(n, period, 0, period / 50)
    filt = orc.generate_filter(period, hw, 0, "both", None)
    x = synth_recording(6, n, 22000, 130, seed=41).astype(dtype)
    for devices in _device_lists():
        y = sharding.filter_host_sharded(filt, x, devices=devices, out_dtype=out_dtype, chunk_samples=150_000)
        ref = orc.filter_data_direct(x, filt)
        assert y.dtype == out_dtype and np.abs(y - ref).max() <= tol * np.abs(ref).max(), devices


def test_filter_host_sharded_blocks_are_views_of_one_large_array():
    """channel blocks are VIEWS of one >= 128 MiB parent array, filtered from two threads at once --
    the parent is page-locked once for the call, no block locks or unlocks pages it shares with its neighbour."""
    period = 22000 / 130 * (1 + 3e-5)
    n = 2_400_000
    hw = orc.default_filter_half_width(n, period, 0, period / 50)
    filt = orc.generate_filter(period, hw, 0, "both", None)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((7, n))  # 134 MB
    assert x.nbytes >= 128 << 20
    y = sharding.filter_host_sharded(filt, x, devices=[0, 0])
    plan = _hip.FilterPlan(filt)
    ref = plan.apply(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    # and again: the first call's lock was released, nothing is left registered
    y2 = sharding.filter_host_sharded(filt, x, devices=[0, 0, 0])
    assert np.abs(y2 - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.float16])
def test_numpy_caller_pipeline_equals_the_one_piece_call(monkeypatch, dtype):
    """The reference's calling convention (ndarray in, fresh ndarray out, parrm.py:835-875) on a host array large
    enough for the overlapped channel-block pipeline (threshold lowered here): same bits as the one-piece call,
    float64 out for any input type, uneven blocks."""
    from pyparrm_amd import parrm as facade

    fs, fa = 22000.0, 130.0
    x = synth_recording(7, 300_000, fs, fa, seed=51).astype(dtype)
    p = PARRM(x, fs, fa, verbose=False)
    p._period = np.float64(fs / fa * (1 + 3e-5))
    p.create_filter()
    monkeypatch.setattr(facade, "_PIPELINE_BYTES", 1 << 60)
    whole = p.filter_data()
    monkeypatch.setattr(facade, "_PIPELINE_BYTES", 1 << 20)
    monkeypatch.setattr(facade, "_PIPELINE_BLOCK_BYTES", x.nbytes // 3)
    piped = p.filter_data()
    assert piped.dtype == np.float64 and piped is not whole and np.array_equal(piped, whole)
    ref = orc.filter_data_direct(x.astype(np.float64), p.filter)
    tol = 1e-10 if dtype == np.float64 else 1e-6
    assert np.abs(piped - ref).max() <= tol * np.abs(ref).max()


def test_fit_error_slices_of_at_most_64_candidates_match_the_whole_grid_bitwise():
    """a slice of <= 64 candidates at K = 41 takes the one-wave-per-quarter solve (+ fit_finish_kernel),
    the whole grid the four-wave form -- 8 ranks x ~48 stage-3 candidates is exactly the 8-GPU strong-scaling shape.
    Same bits required, 256 channels."""
    rng = np.random.default_rng(4)
    n_idx, n_chans = 6000, 256
    idx = np.sort(rng.choice(60_000, size=n_idx, replace=False)).astype(np.int64)
    y = torch.from_numpy(rng.standard_normal((n_idx, n_chans))).cuda()
    d_idx = torch.from_numpy(idx).cuda()
    grid = 169.2358 * (1 + np.linspace(-1e-3, 1e-3, 381))
    for bw in (20, 10, 5):
        whole = _hip.fit_errors(y, d_idx, grid, bw, 1.0)
        for world in (8, 6):
            parts = [_hip.fit_errors(y, d_idx, grid[lo:hi], bw, 1.0, grid_periods=grid.shape[0])
                     for lo, hi in sharding.even_split(grid.shape[0], world)]
            assert max(len(p) for p in parts) <= 64
            assert np.array_equal(np.concatenate(parts), whole), (bw, world)


_TWO_RANK_EXCHANGE = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PARRM_REPO"])
from pyparrm_amd import PARRM, sharding
from pyparrm_amd.synth import synth_recording_exact

torch.cuda.set_device(0)  # both ranks on the one GPU of the box (a multi-GPU node gives each rank its own)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
kind = os.environ["PARRM_TEST_EXCHANGE"]
ex = sharding.IpcExchange.create(dist) if kind == "ipc" else sharding.ShmExchange(dist)
assert ex is not None, "IPC-handle exchange unavailable"
for rows in (5, 70_000, 33):  # (growing pieces: the exchange buffers / segments are re-made once)
    t = torch.arange(rows * 6, dtype=torch.float64, device="cuda").reshape(rows, 6) + 1000.0 * rank
    parts = ex.all_gather(t)
    assert len(parts) == world and all(p.is_cuda for p in parts)
    for r, p in enumerate(parts):
        assert torch.equal(p, torch.arange(rows * 6, dtype=torch.float64, device="cuda").reshape(rows, 6) + 1000.0 * r)
fs, fa = 22000.0, 130.0
x = synth_recording_exact(5, 60_000, fs / fa * (1 + 2e-5), seed=9)
one = PARRM(x, fs, fa, verbose=False)
one.find_period(random_seed=3)
sh = sharding.ShardedPARRM(sharding.shard_recording(x, rank, world), fs, fa, ex, verbose=False)
sh.find_period(random_seed=3)
assert sh.period == one.period, (sh.period, one.period)
sh.create_filter(); one.create_filter()
lo, hi = sharding.channel_shard(5, rank, world)
assert np.array_equal(sh.filter_data(), one.filter_data()[lo:hi])
dist.barrier()
if hasattr(ex, "close"):
    ex.close()
dist.destroy_process_group()
print("TWO_RANK_EXCHANGE_OK", rank)
"""


@pytest.mark.parametrize("kind", ["ipc", "shm"])
def test_process_per_rank_exchanges_without_a_collective_library(kind):
    """The launcher path's two RCCL-free exchanges, two rank PROCESSES (on the one GPU a test box has): device-to-device
    copies through IPC memory handles (`IpcExchange`, bench.py's default) and page-locked shared memory
    (`ShmExchange`); the sharded search on top lands on the single-process period, the blocks on its output."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29580 + (1 if kind == "ipc" else 2)
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   PARRM_REPO=root, PARRM_TEST_EXCHANGE=kind, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANK_EXCHANGE], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"TWO_RANK_EXCHANGE_OK {rank}" in so, so[-1500:] + se[-3000:]
