"""Multi-GPU layout of the PARRM hot path: one process per GPU, channels of ONE recording sharded.

What shards (reference shape of the split in brackets):

* ``filter_data``: every channel row is filtered independently (parrm.py:861-866 broadcasts one
  kernel over the columns) -> contiguous channel blocks, one per rank, no exchange; the N-rank result
  is the concatenation of the rank results, bit-identical to the 1-rank result.
* ``find_period`` statistics pass (parrm.py:274-275): per channel, shards with the rows.
* ``find_period`` candidate grids (parrm.py:445-454 maps over candidates): every rank holds the
  gathered stage matrix of ALL channels (<= 25 001 x C doubles, 51 MB at 256 channels) and evaluates
  a contiguous slice of the grid.  This is the path's one real exchange step, twice per stage:
  replicate the stage columns each rank gathered from its rows, and concatenate the slices' errors
  (<= 80 KB).  The only cross-channel arithmetic of the reference -- the channel mean of
  parrm.py:595-597 -- then happens inside one GPU, in the 1-rank order, and
  ``parrm_fit_errors_slice`` plans a slice as the whole grid would be planned, so the N-rank errors
  are bit-identical to the 1-rank ones and so is the period.
* the Nelder-Mead refinements (parrm.py:510-517, :545-550) are a chain of dependent evaluations:
  every rank runs them redundantly on its replica (no exchange, identical results).

The exchange goes through an :class:`Exchange` object: ``TorchExchange`` rides on the process
group the launcher created (``all_gather``; RCCL over xGMI when the group's backend is nccl, host
memory when it is gloo), ``ThreadExchange`` runs N ranks as threads of one process (tests).

``timed_steps`` is the benchmark's timing protocol (barrier + MAX over ranks).
"""

from __future__ import annotations

import threading
import time

import numpy as np

from . import _hip
from .parrm import PARRM, _STAGE_LAMBDA


def even_split(n_items: int, n_parts: int) -> list[tuple[int, int]]:
    """Contiguous ``[lo, hi)`` ranges covering ``range(n_items)``, sizes differing by at most one
    (the first ``n_items % n_parts`` parts take the extra item)."""
    if n_parts <= 0:
        raise ValueError("`n_parts` must be > 0.")
    if n_items < 0:
        raise ValueError("`n_items` must be >= 0.")
    base, extra = divmod(n_items, n_parts)
    out, lo = [], 0
    for part in range(n_parts):
        hi = lo + base + (1 if part < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def channel_shard(n_chans: int, rank: int, world_size: int) -> tuple[int, int]:
    """Channel block ``[lo, hi)`` of ``rank`` (may be empty when ``world_size > n_chans``)."""
    _check_rank(rank, world_size)
    return even_split(n_chans, world_size)[rank]


def candidate_slice(n_candidates: int, rank: int, world_size: int) -> tuple[int, int]:
    """Slice ``[lo, hi)`` of a period grid evaluated by ``rank``."""
    _check_rank(rank, world_size)
    return even_split(n_candidates, world_size)[rank]


def shard_recording(data, rank: int, world_size: int):
    """View of the channel block of ``rank`` (NumPy array or torch tensor; no copy)."""
    lo, hi = channel_shard(data.shape[0], rank, world_size)
    return data[lo:hi]


def _check_rank(rank: int, world_size: int) -> None:
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("`rank` must lie in the range [0, world_size).")


# ------------------------------------------------------------------------------ exchange
class Exchange:
    """How the ranks of a sharded search swap their pieces.  ``all_gather(t)`` returns the list of
    every rank's tensor, in rank order; all ranks pass tensors of one shape and dtype."""

    rank = 0
    world_size = 1

    def all_gather(self, tensor):
        return [tensor]


class TorchExchange(Exchange):
    """``torch.distributed`` process group (one process per GPU).  With the nccl backend (= RCCL) the
    pieces travel device to device over xGMI; with gloo they are staged through host memory."""

    def __init__(self, dist, group=None, via_host: bool = False):
        """``via_host=True`` keeps RCCL out of the data path altogether: the pieces are staged through
        host memory over a gloo group created next to the launcher's (slower: the 51 MB stage matrix
        of a 256-channel recording then crosses PCIe and host sockets instead of xGMI)."""
        if via_host and dist.get_backend(group) == "nccl":
            group = dist.new_group(backend="gloo")
        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self._via_host = dist.get_backend(group) != "nccl"

    def all_gather(self, tensor):
        import torch

        src = tensor.contiguous()
        if self._via_host and src.is_cuda:
            host = src.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world_size)]
            self._dist.all_gather(parts, host, group=self._group)
            return [part.to(src.device) for part in parts]
        parts = [torch.empty_like(src) for _ in range(self.world_size)]
        self._dist.all_gather(parts, src, group=self._group)
        return parts


class ThreadExchange(Exchange):
    """N ranks as N threads of one process (one GPU): ``ThreadExchange.group(n)`` returns the ranks'
    exchange objects.  Used to check on a single device that a sharded run reproduces the 1-rank run."""

    def __init__(self, rank, world_size, slots, barrier):
        self.rank, self.world_size = rank, world_size
        self._slots, self._barrier = slots, barrier

    @classmethod
    def group(cls, world_size: int):
        slots, barrier = [None] * world_size, threading.Barrier(world_size)
        return [cls(rank, world_size, slots, barrier) for rank in range(world_size)]

    def all_gather(self, tensor):
        self._slots[self.rank] = tensor
        self._barrier.wait()
        parts = list(self._slots)
        self._barrier.wait()  # nobody overwrites a slot before every rank has read it
        return parts


# ------------------------------------------------------------------------------ sharded facade
class ShardedPARRM(PARRM):
    """``PARRM`` for ONE recording whose channels are spread over the ranks of an ``Exchange``.

    Each rank constructs it with ITS channel block (``shard_recording(data, rank, world)`` of the
    same recording) and calls the usual ``find_period`` -> ``create_filter`` -> ``filter_data`` with the
    same arguments on every rank.  ``period`` is the period of the WHOLE recording -- bit-identical
    on every rank and to what one ``PARRM`` on the unsharded recording returns; ``filter_data()``
    returns this rank's block of the filtered recording.
    """

    def __init__(self, data, sampling_freq, artefact_freq, exchange: Exchange, verbose=True) -> None:
        super().__init__(data, sampling_freq, artefact_freq, verbose)
        if not isinstance(exchange, Exchange):
            raise TypeError("`exchange` must be a sharding.Exchange.")
        self._exchange = exchange
        self._block_sizes = None  # channels held by every rank (learnt at the first exchange)

    def __deepcopy__(self, memo):
        clone = super().__deepcopy__(memo)
        clone._exchange = self._exchange  # a communicator is shared, not copied
        return clone

    def _learn_block_sizes(self, device) -> None:
        torch = _hip.require_gpu()
        if self._block_sizes is None:
            mine = torch.tensor([self._n_chans], dtype=torch.int64, device=device)
            self._block_sizes = [int(t.item()) for t in self._exchange.all_gather(mine)]
            if min(self._block_sizes) < 1:
                raise ValueError("every rank of a sharded recording needs at least one channel")

    def _total_chans(self, x):
        """filter_data cuts the time axis as for the WHOLE recording (``parrm_filter_apply_block``), so a
        block's output is bit-identical to the same rows of the unsharded result.  Applies to the
        object's own block; any other array is filtered as a recording of its own."""
        if self._exchange.world_size == 1 or x.shape[0] != self._n_chans:
            return int(x.shape[0])
        self._learn_block_sizes(x.device)
        return sum(self._block_sizes)

    # stage columns of ALL channels on every rank
    def _stage_matrix(self, indices: np.ndarray):
        torch = _hip.require_gpu()
        y_local, d_idx = super()._stage_matrix(indices)
        ex = self._exchange
        if ex.world_size == 1:
            return y_local, d_idx
        self._learn_block_sizes(y_local.device)
        widest = max(self._block_sizes)
        padded = torch.zeros((y_local.shape[0], widest), dtype=torch.float64, device=y_local.device)
        padded[:, : self._n_chans] = y_local
        blocks = ex.all_gather(padded)
        total = sum(self._block_sizes)
        ld = (total + 3) // 4 * 4  # whole column quads, zero-filled: the layout gather_standardise writes
        y = torch.zeros((y_local.shape[0], ld), dtype=torch.float64, device=y_local.device)
        col = 0
        for block, width in zip(blocks, self._block_sizes):
            y[:, col:col + width] = block[:, :width]
            col += width
        return y[:, :total], d_idx

    # this rank's slice of the grid, then the concatenation of all slices
    def _grid_errors(self, y, d_idx, grid: np.ndarray, bandwidth: int, workspace) -> np.ndarray:
        torch = _hip.require_gpu()
        ex = self._exchange
        if ex.world_size == 1:
            return super()._grid_errors(y, d_idx, grid, bandwidth, workspace)
        n_grid = grid.shape[0]
        lo, hi = candidate_slice(n_grid, ex.rank, ex.world_size)
        mine = np.empty(0)
        if hi > lo:
            mine = _hip.fit_errors(y, d_idx, grid[lo:hi], bandwidth, _STAGE_LAMBDA, workspace, grid_periods=n_grid)
        widest = (n_grid + ex.world_size - 1) // ex.world_size
        padded = np.zeros(widest)
        padded[: hi - lo] = mine
        parts = ex.all_gather(torch.from_numpy(padded).to(y.device))
        out = np.empty(n_grid)
        for rank, part in enumerate(parts):
            rlo, rhi = candidate_slice(n_grid, rank, ex.world_size)
            out[rlo:rhi] = part[: rhi - rlo].cpu().numpy()
        return out

    def __repr__(self) -> str:
        ex = self._exchange
        return (f"ShardedPARRM object | rank {ex.rank} of {ex.world_size} | Data: ({self._n_chans} channels x "
                f"{self._n_samples} times) | Period: {self._period:.4f}")


def timed_steps(step, n_steps: int, n_warmup: int, dist=None, sync=None) -> float:
    """Benchmark protocol shared by ``bench.py`` and the tests: ``n_warmup`` untimed calls of
    ``step()``, then exactly ``n_steps`` timed calls bracketed by ``sync()`` + barrier on both
    sides; returns the MAX over ranks of the elapsed seconds.

    ``dist`` is an initialised ``torch.distributed`` module (or None for one process); ``sync``
    is e.g. ``torch.cuda.synchronize``."""

    def fence():
        if sync is not None:
            sync()
        if dist is not None:
            dist.barrier()
        if sync is not None:
            sync()

    for _ in range(n_warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        backend = dist.get_backend()
        device = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
