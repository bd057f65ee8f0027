"""N>1 path on CPU: world_size-2 ``gloo`` processes.

``test_two_rank_gloo``: the shard arithmetic (index helpers, timing protocol) with the oracle doing the
work inside each shard.  ``test_sharded_parrm_two_rank_gloo``: the PRODUCT's sharded search
(``sharding.ShardedPARRM`` + ``TorchExchange`` on a gloo group) end to end -- channel blocks, the
stage-matrix replication, candidate slices, the replicated Nelder-Mead, per-rank filtering -- with the
device entry points of ``pyparrm_amd._hip`` replaced by the oracle's arithmetic (there is no GPU here);
both ranks must land on exactly the period the oracle finds on the unsharded recording."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pyparrm_amd import sharding  # noqa: E402


def test_even_split_properties():
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8, 16):
            parts = sharding.even_split(n, w)
            assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.even_split(256, 8) == [(32 * i, 32 * i + 32) for i in range(8)]
    with pytest.raises(ValueError):
        sharding.channel_shard(4, 2, 2)
    with pytest.raises(ValueError):
        sharding.even_split(4, 0)
    x = np.arange(12).reshape(6, 2)
    assert np.shares_memory(sharding.shard_recording(x, 1, 2), x)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from oracle import parrm_oracle as orc
    from pyparrm_amd import sharding as sh
    from pyparrm_amd.synth import synth_recording

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fs, fa = 22000.0, 130.0
        x = synth_recording(5, 20000, fs, fa, seed=3)  # 5 channels: uneven split 3 + 2
        filt = orc.generate_filter(169.23584615384616, 2372, 0, "both", None)

        # (1) channel-sharded filter_data: rank results concatenate to the unsharded result
        lo, hi = sh.channel_shard(x.shape[0], rank, world)
        mine = orc.filter_data_direct(x[lo:hi], filt)
        padded = np.zeros((3, x.shape[1]))
        padded[: hi - lo] = mine
        gathered = [torch.zeros((3, x.shape[1]), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(padded))  # verification only
        parts = [gathered[r][: sh.channel_shard(5, r, world)[1] - sh.channel_shard(5, r, world)[0]].numpy()
                 for r in range(world)]
        whole = orc.filter_data_direct(x, filt)
        assert np.array_equal(np.concatenate(parts), whole)

        # (2) candidate-sliced period grid with a replicated stage matrix
        std = orc.standardise_data(x, 3.0)
        idx = np.arange(7500, 12501)
        grid = orc.possible_periods((fs / fa,), 1)[::16]
        glo, ghi = sh.candidate_slice(grid.shape[0], rank, world)
        errs = orc.grid_errors(grid[glo:ghi], std, idx, 5, 1.0)
        buf = np.full(grid.shape[0], np.nan)
        buf[glo:ghi] = errs
        t = torch.from_numpy(np.nan_to_num(buf, nan=0.0))
        dist.all_reduce(t)  # disjoint slices: the sum is the concatenation (verification only)
        full = orc.grid_errors(grid, std, idx, 5, 1.0)
        assert np.array_equal(t.numpy(), full)

        # (3) the benchmark's timing protocol: barrier on both sides, MAX over ranks
        calls = []
        elapsed = sh.timed_steps(lambda: calls.append(1) or __import__("time").sleep(0.01 * (rank + 1)),
                                 n_steps=3, n_warmup=1, dist=dist)
        assert len(calls) == 4
        assert elapsed >= 3 * 0.01 * world - 1e-3  # every rank reports the slowest rank's time
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    out = tmp_path / "ok.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _install_oracle_device(torch, orc):
    """Replace the device entry points the façade calls by oracle arithmetic on CPU tensors."""
    from pyparrm_amd import _hip, parrm as facade

    class Plan:
        def __init__(self, filt, device=None):
            self.filt, self.device = filt, device

        def apply(self, x, out=None, out_dtype=None, total_chans=None):
            return torch.from_numpy(orc.filter_data_direct(x.numpy(), self.filt))

    def gather(x, idx, scale, ob):
        d = np.diff(x.numpy().astype(np.float64), axis=1)[:, idx.numpy()]
        return torch.from_numpy(np.ascontiguousarray(np.clip(d / scale.numpy()[:, None], -ob, ob).T))

    def fit_errors(y, idx, periods, bw, lam, workspace=None, grid_periods=0):
        rows = np.ascontiguousarray(y.numpy().T)
        return np.array([orc.fit_error_gathered(p, rows, rows.shape[0], idx.numpy(), bw, lam)
                         for p in np.asarray(periods, dtype=np.float64).reshape(-1)])

    _hip.require_gpu = lambda: torch
    _hip.absdiff_mean = lambda x: torch.from_numpy(np.abs(np.diff(x.numpy(), axis=1)).mean(axis=1))
    _hip.gather_standardise = gather
    _hip.fit_errors = fit_errors
    _hip.FitWorkspace = lambda: None
    facade._NM_IN_PYTHON = True  # the refinement step by step through the stand-in objective (not the library's own loop)
    _hip.FilterPlan = Plan
    _hip.to_host_numpy = lambda t: t.numpy()
    _hip.upload_indices = lambda indices, device: torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int64))
    facade.PARRM._device_recording = lambda self, data=None: torch.from_numpy(
        np.ascontiguousarray(self._data if data is None else data))
    facade.PARRM._plan_for = lambda self, device: Plan(self._filter)


def _sharded_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from oracle import parrm_oracle as orc
    from pyparrm_amd import sharding as sh
    from pyparrm_amd.synth import synth_recording_exact

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _install_oracle_device(torch, orc)
        fs, fa = 1000.0, 130.0
        x = synth_recording_exact(3, 2600, fs / fa * (1 + 4e-4), seed=77)  # 3 channels: blocks of 2 + 1
        mine = sh.shard_recording(x, rank, world)
        p = sh.ShardedPARRM(mine, fs, fa, sh.TorchExchange(dist), verbose=False)
        p.find_period(random_seed=5)
        whole = float(orc.find_period(x, fs, fa, random_seed=5))
        assert float(p.period) == whole, (p.period, whole)
        assert len(p._trace[0]["errors"]) == len(p._trace[0]["grid"])  # every rank holds the whole grid's errors
        p.create_filter()
        y = p.filter_data()
        lo, hi = sh.channel_shard(3, rank, world)
        assert np.array_equal(y, orc.filter_data_direct(x, p.filter)[lo:hi])
        periods = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(periods, torch.tensor([float(p.period)], dtype=torch.float64))  # verification only
        assert all(float(t) == float(p.period) for t in periods)
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_parrm_two_rank_gloo(tmp_path):
    out = tmp_path / "ok.txt"
    mp.spawn(_sharded_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _shm_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from oracle import parrm_oracle as orc
    from pyparrm_amd import sharding as sh
    from pyparrm_amd.synth import synth_recording_exact

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = sh.ShmExchange(dist, tag=f"parrmtest{port}")
    try:
        # the exchange itself: pieces of every rank, in rank order, through growing segments
        for rows in (3, 50_000, 7):
            t = torch.arange(rows * 4, dtype=torch.float64).reshape(rows, 4) + 1000.0 * rank
            parts = ex.all_gather(t)
            assert len(parts) == world
            for r, part in enumerate(parts):
                assert torch.equal(part, torch.arange(rows * 4, dtype=torch.float64).reshape(rows, 4) + 1000.0 * r)
        # the product's sharded search on top of it
        _install_oracle_device(torch, orc)
        fs, fa = 1000.0, 130.0
        x = synth_recording_exact(3, 2600, fs / fa * (1 + 4e-4), seed=77)
        p = sh.ShardedPARRM(sh.shard_recording(x, rank, world), fs, fa, ex, verbose=False)
        p.find_period(random_seed=5)
        assert float(p.period) == float(orc.find_period(x, fs, fa, random_seed=5))
        p.create_filter()
        lo, hi = sh.channel_shard(3, rank, world)
        assert np.array_equal(p.filter_data(), orc.filter_data_direct(x, p.filter)[lo:hi])
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        ex.close()
        dist.destroy_process_group()


def test_sharded_parrm_over_shared_memory_exchange(tmp_path):
    """``ShmExchange``: one process per rank, the pieces staged through POSIX shared memory (no collective
    library in the data path; ``torch.distributed`` -- gloo here -- only synchronises)."""
    out = tmp_path / "ok.txt"
    mp.spawn(_shm_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _multi_device_in_subprocess(_rank, out_path):
    sys.path.insert(0, ROOT)
    from oracle import parrm_oracle as orc
    from pyparrm_amd import sharding as sh
    from pyparrm_amd.synth import synth_recording_exact

    _install_oracle_device(torch, orc)
    # no GPU here: the thread-ranks' device binding and peer copies become no-ops on CPU tensors
    torch.cuda.set_device = lambda d: None
    sh.DeviceExchange.all_gather = sh.ThreadExchange.all_gather
    fs, fa = 1000.0, 130.0
    x = synth_recording_exact(5, 2600, fs / fa * (1 + 4e-4), seed=78)
    p = sh.MultiDevicePARRM(x, fs, fa, devices=[0, 0, 0], verbose=False)
    p.find_period(random_seed=5)
    assert float(p.period) == float(orc.find_period(x, fs, fa, random_seed=5))
    with pytest.raises(ValueError):
        sh.MultiDevicePARRM(x, fs, fa, devices=[0, 0], verbose=False).filter_data()  # the reference's call-order rule
    p.create_filter()
    y = p.filter_data()
    assert y.shape == x.shape and np.array_equal(y, orc.filter_data_direct(x, p.filter))
    other = x[:4, :1500] * 2.0
    assert np.array_equal(p.filter_data(other), orc.filter_data_direct(other, p.filter))
    assert isinstance(p.settings, dict) and p.filtered_data is not None
    # an error on one rank comes back to the caller and leaves the object usable
    with pytest.raises(ValueError):
        p.find_period(outlier_boundary=-1.0)
    p.find_period(random_seed=5)
    open(out_path, "w").write("ok")


def test_multi_device_facade_threads_with_oracle_device(tmp_path):
    """``MultiDevicePARRM`` (one process, one host thread per device): the reference's surface, NumPy in and out,
    three thread-ranks over blocks of 2 + 2 + 1 channels; the oracle stands in for the device (in a child process:
    the stand-ins are monkey-patches of the binding)."""
    out = tmp_path / "ok.txt"
    mp.spawn(_multi_device_in_subprocess, args=(str(out),), nprocs=1, join=True)
    assert out.read_text() == "ok"
