"""N>1 path on CPU: world_size-2 ``gloo`` processes.  The shard planning is the product code;
the arithmetic inside each shard is done by the oracle here (no GPU), and the collectives in this
file exist only to compare the ranks' results -- the product path has none."""

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
