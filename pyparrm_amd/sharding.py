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

The exchange goes through an :class:`Exchange` object:

* ``DeviceExchange`` -- N ranks as N host threads of ONE process, one device each, pieces moved by peer copies:
  what :class:`MultiDevicePARRM` (the single-process, NumPy-in / NumPy-out facade over all the GPUs of a node;
  SURVEY.md section 7 step 6: "one host thread + stream per device, no collectives") is built on;
* ``IpcExchange`` -- one process per GPU (a launcher such as ``torch.distributed.run``), pieces copied device to
  device through IPC memory handles (peer copies over xGMI): no RCCL anywhere in the data path (BASELINE.json's
  north_star), no host staging; ``torch.distributed`` only carries the handles and the barrier.  ``bench.py
  --gpus N`` uses it by default;
* ``ShmExchange`` -- the same processes, pieces staged through page-locked POSIX shared memory: the fallback where
  IPC handles cannot be used;
* ``TorchExchange`` -- the same processes, pieces by ``all_gather`` on the launcher's process group (RCCL over xGMI
  when its backend is nccl, host memory when it is gloo): the fallback when shared memory is not available;
* ``ThreadExchange`` -- N thread-ranks on one device (tests).

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


class DeviceExchange(ThreadExchange):
    """N ranks as N host threads of one process, rank r bound to ``devices[r]`` (the same device may appear more
    than once: rehearsals on one GPU).  ``all_gather`` hands every rank the other ranks' tensors as copies on ITS
    device -- peer-to-peer copies over xGMI where the devices can reach each other, staged by the runtime otherwise;
    no collective library is involved."""

    def __init__(self, rank, world_size, slots, barrier, device):
        super().__init__(rank, world_size, slots, barrier)
        self.device = device

    @classmethod
    def group(cls, devices):
        devices = list(devices)
        slots, barrier = [None] * len(devices), threading.Barrier(len(devices))
        return [cls(rank, len(devices), slots, barrier, dev) for rank, dev in enumerate(devices)]

    def all_gather(self, tensor):
        import torch

        # the producer's work on its stream must be complete before another thread's stream reads the tensor
        if tensor.is_cuda:
            torch.cuda.current_stream(tensor.device).synchronize()
        self._slots[self.rank] = tensor
        self._barrier.wait()
        mine = torch.device("cuda", self.device)
        parts = [t if (not t.is_cuda or t.device == mine) else t.to(mine) for t in self._slots]
        if tensor.is_cuda:
            torch.cuda.current_stream(mine).synchronize()  # the copies have read their sources
        self._barrier.wait()  # nobody overwrites a slot before every rank has read it
        return parts


class _AttachedSegment:
    """Another rank's shared-memory segment, mapped read/write WITHOUT ``multiprocessing.shared_memory``: before
    Python 3.13 that class registers attached segments with the resource tracker too, which then unlinks -- or warns
    about -- segments this process never owned (and, where the ranks share one tracker, the owner's ``unlink`` and
    an attacher's ``unregister`` of the same name leave a ``KeyError`` traceback in the tracker at teardown)."""

    def __init__(self, name: str):
        import mmap
        import os

        import _posixshmem

        fd = _posixshmem.shm_open("/" + name.lstrip("/"), os.O_RDWR, mode=0o600)
        try:
            self.size = os.fstat(fd).st_size
            self._mmap = mmap.mmap(fd, self.size)
        finally:
            os.close(fd)
        self.buf = memoryview(self._mmap)

    def close(self):
        if self._mmap is not None:
            self.buf.release()
            self._mmap.close()
            self._mmap = None


class ShmExchange(Exchange):
    """One process per GPU, pieces staged through POSIX shared memory (``/dev/shm``): every rank owns one segment,
    writes its piece there (device -> host copy straight into the mapping), and reads the other ranks' segments
    (host -> device) between two barriers of a gloo group.  Nothing of RCCL is in the data path; the 51 MB stage
    matrix of a 256-channel recording crosses PCIe twice (~2 ms) instead of xGMI once.

    ``dist``: an initialised ``torch.distributed``; a gloo group for the barriers is created next to the default
    group when that one is nccl."""

    def __init__(self, dist, tag: str | None = None):
        import os

        self._dist = dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self._group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None
        self._tag = tag or f"parrm{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self._mine = None          # (SharedMemory, capacity)
        self._theirs = {}          # rank -> (SharedMemory, capacity)
        self._staging = None       # page-locked staging buffer (uint8 tensor; False: none to be had)
        self._seq = 0

    def _barrier(self):
        self._dist.barrier(group=self._group)

    def _segment(self, nbytes: int):
        from multiprocessing import shared_memory

        if self._mine is None or self._mine[1] < nbytes:
            if self._mine is not None:
                self._mine[0].close()
                self._mine[0].unlink()
            cap = max(nbytes, 1 << 20)
            name = f"{self._tag}_r{self.rank}_g{self._seq}"
            try:
                seg = shared_memory.SharedMemory(name=name, create=True, size=cap)
            except FileExistsError:  # left behind by an earlier job that died with the same rendezvous port
                stale = shared_memory.SharedMemory(name=name)
                stale.close()
                stale.unlink()
                seg = shared_memory.SharedMemory(name=name, create=True, size=cap)
            self._mine = (seg, cap, name)
        return self._mine

    def _stage(self, nbytes: int, device_side: bool):
        """A page-locked staging buffer of this exchange (torch's pinned allocator, i.e. an ALLOCATION, grow-only):
        device <-> staging runs at the link's rate, staging <-> mapping is a host memcpy.  Round 3 page-locked the
        mappings themselves with ``hipHostRegister``; a segment that grows is unmapped while registered and its
        address range can come back with the next mapping and be registered again -- the lock / unlock / lock
        pattern ``profiles/r03_host_register_fault.txt`` ties to GPU page faults on this runtime.
        Nothing in this class registers host memory any more."""
        import torch

        if not device_side:
            return None
        if self._staging is None or self._staging.numel() < nbytes:
            try:
                self._staging = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
            except RuntimeError:  # no page-locked memory to be had: the pageable path is merely slower
                self._staging = False
        return self._staging if self._staging is not False and self._staging.numel() >= nbytes else None

    def all_gather(self, tensor):
        import torch
        from multiprocessing import shared_memory

        src = tensor.contiguous()
        nbytes = src.numel() * src.element_size()
        self._seq += 1
        seg = self._segment(nbytes)
        host = torch.frombuffer(seg[0].buf, dtype=src.dtype, count=src.numel()).view(src.shape)
        stage = self._stage(nbytes, src.is_cuda)
        if stage is not None:  # device -> page-locked staging -> shared memory
            staged = stage[:nbytes].view(src.dtype).view(src.shape)
            staged.copy_(src)  # (blocking: the destination is host memory)
            host.copy_(staged)
        else:
            host.copy_(src)  # device (pageable path) or host -> shared memory, blocking
        # every rank tells the others which segment holds its piece of this round (names change when one grows)
        names = [None] * self.world_size
        self._dist.all_gather_object(names, seg[2], group=self._group)  # doubles as the "everybody has written" barrier
        parts = []
        for rank, name in enumerate(names):
            if rank == self.rank:
                parts.append(src)
                continue
            cur = self._theirs.get(rank)
            if cur is None or cur[1] != name:
                if cur is not None:
                    cur[0].close()
                cur = (_AttachedSegment(name), name)
                self._theirs[rank] = cur
            view = torch.frombuffer(cur[0].buf, dtype=src.dtype, count=src.numel()).view(src.shape)
            if stage is not None:  # shared memory -> page-locked staging -> device
                staged = stage[:nbytes].view(src.dtype).view(src.shape)
                staged.copy_(view)
                parts.append(staged.to(src.device))  # (blocking: the staging buffer is free for the next piece)
            else:
                parts.append(view.to(src.device) if src.is_cuda else view.clone())
        self._barrier()  # nobody rewrites its segment before every rank has read it
        return parts

    def close(self):
        self._staging = None
        for seg in self._theirs.values():
            seg[0].close()
        self._theirs = {}
        if self._mine is not None:
            self._mine[0].close()
            try:
                self._mine[0].unlink()
            except FileNotFoundError:
                pass
            self._mine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IpcExchange(Exchange):
    """One process per GPU, pieces moved DEVICE TO DEVICE without a collective library: every rank publishes an IPC
    handle of its piece (``torch.multiprocessing.reductions.reduce_tensor``: hipIpcGetMemHandle + an interprocess
    event), the handles travel as small pickled objects over a gloo group, and every rank copies the other ranks'
    pieces straight into its own memory (peer copies over xGMI).  No RCCL anywhere in the data path and no host
    staging; ``torch.distributed`` only carries the handles and the barrier.  Needs ``HSA_ENABLE_IPC_MODE_LEGACY=0``
    on this platform (exported by the image) and peer access between the devices.

    ``IpcExchange.create(dist)`` probes the mechanism on every rank and returns None -- on EVERY rank -- if any rank
    cannot use it; callers then fall back to :class:`ShmExchange`."""

    def __init__(self, dist, group):
        self._dist, self._group = dist, group
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self._mine = None    # this rank's exchange buffer (uint8 device tensor), mapped by every peer
        self._peers = {}     # rank -> that rank's buffer as mapped into this process

    @classmethod
    def create(cls, dist):
        """Probe the mechanism in PHASES THAT EVERY RANK ALWAYS EXECUTES -- allocate, exchange handles, exchange "my
        mappings opened" flags, copy, exchange "my copies were right" flags -- so that a failure on one rank (a pair
        without peer access, a handle that does not open) never makes it skip a collective the others are waiting
        in: every rank learns the verdict in the same call and all of them return None together (the
        earlier form went from a failed copy straight to the flags while its peers sat in a barrier)."""
        import torch
        from torch.multiprocessing.reductions import reduce_tensor

        group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None
        ex = cls(dist, group)
        world, rank = ex.world_size, ex.rank

        def agree(ok: bool) -> bool:
            flags = [None] * world
            dist.all_gather_object(flags, bool(ok), group=group)
            return all(flags)

        # phase 1: this rank's buffer and its handle (None: could not be made)
        handle, ok = None, True
        try:
            ex._mine = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
            ex._mine[:8192].view(torch.float64).fill_(float(rank))
            torch.cuda.current_stream().synchronize()
            handle = reduce_tensor(ex._mine)
        except Exception:  # noqa: BLE001 -- any failure means "not usable here"
            ok = False
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
        # phase 2: open the peers' buffers
        if ok and all(h is not None for h in handles):
            try:
                for r, (rebuild, args) in enumerate(handles):
                    if r != rank:
                        ex._peers[r] = rebuild(*args)
            except Exception:  # noqa: BLE001
                ok = False
        else:
            ok = False
        if not agree(ok):
            ex._mine, ex._peers = None, {}
            return None
        # phase 3: read every peer's pattern through the mapping
        try:
            for r, remote in ex._peers.items():
                got = remote[:8192].view(torch.float64).to("cuda", copy=True)
                ok = ok and float(got[0].item()) == float(r) and float(got[-1].item()) == float(r)
            torch.cuda.current_stream().synchronize()
        except Exception:  # noqa: BLE001
            ok = False
        if not agree(ok):
            ex._mine, ex._peers = None, {}
            return None
        return ex

    def _ensure(self, nbytes: int, device):
        """Exchange buffers of at least `nbytes` on every rank (all ranks pass pieces of one size, so all grow
        together): allocated once, their IPC handles exchanged once, the mappings kept -- an `all_gather` is then
        copies and two barriers, no handle traffic."""
        import torch
        from torch.multiprocessing.reductions import reduce_tensor

        if self._mine is not None and self._mine.numel() >= nbytes:
            return
        self._peers = {}
        self._mine = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()
        handles = [None] * self.world_size
        self._dist.all_gather_object(handles, reduce_tensor(self._mine), group=self._group)
        for rank, (rebuild, args) in enumerate(handles):
            if rank != self.rank:
                self._peers[rank] = rebuild(*args)

    def all_gather(self, tensor):
        import torch

        src = tensor.contiguous()
        if not src.is_cuda:
            parts = [torch.empty_like(src) for _ in range(self.world_size)]
            self._dist.all_gather(parts, src, group=self._group)
            return parts
        nbytes = src.numel() * src.element_size()
        self._ensure(nbytes, src.device)
        mine = self._mine[:nbytes].view(src.dtype).view(src.shape)
        mine.copy_(src)
        torch.cuda.current_stream(src.device).synchronize()
        self._dist.barrier(group=self._group)  # every rank's piece is in its buffer
        parts = []
        for rank in range(self.world_size):
            if rank == self.rank:
                parts.append(src)
            else:
                remote = self._peers[rank][:nbytes].view(src.dtype).view(src.shape)
                parts.append(remote.to(src.device, copy=True))  # device-to-device copy into our own memory
        torch.cuda.current_stream(src.device).synchronize()
        self._dist.barrier(group=self._group)  # nobody rewrites its buffer before every rank has copied it
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


class MultiDevicePARRM:
    """The ``PARRM`` surface for ONE recording spread over several GPUs of THIS process: NumPy array in, NumPy
    array out, no launcher, no process group, no collective library.

        p = MultiDevicePARRM(data, sampling_freq, artefact_freq)            # all visible devices
        p.find_period(); p.create_filter(); clean = p.filter_data()

    One host thread per device drives a :class:`ShardedPARRM` on that device's channel block (SURVEY.md section 7
    step 6); the threads meet in :class:`DeviceExchange` (peer copies) twice per search stage.  ``period`` and the
    filtered recording are bit-identical to a single-device ``PARRM`` on the same array (the candidate grids are
    planned as whole grids, the time axis is cut as for the whole recording).

    ``devices``: device indices, one rank each (default: every visible device; an index may repeat -- rehearsals
    of the N-rank path on one GPU).  A recording with fewer channels than devices uses as many devices as channels."""

    def __init__(self, data, sampling_freq, artefact_freq, devices=None, verbose=True) -> None:
        torch = _hip.require_gpu()
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("`devices` must name at least one GPU.")
        probe = PARRM(data, sampling_freq, artefact_freq, verbose)  # the reference's checks, in its order
        devices = devices[: max(1, min(len(devices), probe._n_chans))]
        self._devices = devices
        self._data = data
        self._verbose = verbose
        exchanges = DeviceExchange.group(devices)
        blocks = [shard_recording(data, r, len(devices)) for r in range(len(devices))]
        self._ranks = [ShardedPARRM(block, sampling_freq, artefact_freq, ex, verbose and r == 0)
                       for r, (block, ex) in enumerate(zip(blocks, exchanges))]
        self._filtered_data = None

    @classmethod
    def from_blocks(cls, blocks, sampling_freq, artefact_freq, verbose=True):
        """The same for a recording that is ALREADY resident: ``blocks`` are CUDA tensors, the consecutive channel
        blocks of one recording, one per device (block r on ``blocks[r].device``).  ``filter_data()`` then returns the
        list of filtered blocks, each on its device -- nothing crosses PCIe."""
        self = cls.__new__(cls)
        blocks = list(blocks)
        if not blocks or any(not getattr(b, "is_cuda", False) or b.ndim != 2 for b in blocks):
            raise TypeError("`blocks` must be a non-empty sequence of 2D CUDA tensors.")
        self._devices = [int(b.device.index) for b in blocks]
        self._data = blocks
        self._verbose = verbose
        exchanges = DeviceExchange.group(self._devices)
        self._ranks = [ShardedPARRM(b, sampling_freq, artefact_freq, ex, verbose and r == 0)
                       for r, (b, ex) in enumerate(zip(blocks, exchanges))]
        self._filtered_data = None
        return self

    @property
    def devices(self):
        return list(self._devices)

    def _on_every_rank(self, call):
        """``call(rank_object)`` on every rank, each in a thread of its own with its device current; the first
        exception (in rank order) is re-raised here after every thread has finished or aborted."""
        import torch

        results, errors = [None] * len(self._ranks), [None] * len(self._ranks)
        barrier = self._ranks[0]._exchange._barrier

        def work(r):
            try:
                torch.cuda.set_device(self._devices[r])
                results[r] = call(self._ranks[r])
            except BaseException as exc:  # noqa: BLE001 -- re-raised in the caller's thread
                errors[r] = exc
                barrier.abort()  # ranks waiting in an exchange must not wait for this one forever

        if len(self._ranks) == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(r,), name=f"parrm-rank{r}") for r in range(len(self._ranks))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real or any(errors):
            barrier.reset()
            raise (real or [e for e in errors if e is not None])[0]
        return results

    def find_period(self, search_samples=None, assumed_periods=None, outlier_boundary=3.0, random_seed=None,
                    n_jobs=1) -> None:
        """parrm.py:148-194 on the whole recording; every rank ends with the same period."""
        self._filtered_data = None
        self._on_every_rank(lambda p: p.find_period(search_samples, assumed_periods, outlier_boundary, random_seed, n_jobs))

    def create_filter(self, filter_half_width=None, omit_n_samples=0, filter_direction="both", period_half_width=None) -> None:
        """parrm.py:689-737 (host arithmetic, identical on every rank)."""
        self._filtered_data = None
        for p in self._ranks:
            p.create_filter(filter_half_width, omit_n_samples, filter_direction, period_half_width)

    def filter_data(self, data=None) -> np.ndarray:
        """parrm.py:835-875: every device filters its channel block; returns the whole filtered recording."""
        if data is None and isinstance(self._data, list):  # resident blocks: results stay on their devices
            self._filtered_data = self._on_every_rank(lambda p: p.filter_data())
            return self._filtered_data
        if data is None:
            blocks = self._on_every_rank(lambda p: np.asarray(p.filter_data()))
        else:
            probe = self._ranks[0]._check_sort_filter_data_inputs(data)  # the reference's checks
            n = len(self._ranks)
            parts = [shard_recording(probe, r, n) for r in range(n)]
            total = int(probe.shape[0])

            def one(p):
                r = p._exchange.rank
                if parts[r].shape[0] == 0:
                    return np.empty((0, probe.shape[1]))
                x = p._device_recording(parts[r])
                try:
                    y = p._plan_for(x.device).apply(x, total_chans=total)
                    return _hip.to_host_numpy(y)
                finally:
                    p._end_call()

            if self._ranks[0]._filter is None:
                raise ValueError("The filter has not yet been created. The `create_filter` method must be called first.")
            blocks = self._on_every_rank(one)
        self._filtered_data = np.concatenate(blocks, axis=0)
        return self._filtered_data

    # the reference's read-only properties (parrm.py:888-936)
    @property
    def data(self):
        return self._data

    @property
    def period(self):
        return self._ranks[0].period

    @property
    def filter(self):
        return self._ranks[0].filter

    @property
    def filtered_data(self):
        if self._filtered_data is None:
            raise AttributeError("No data has been filtered, so the filtered data cannot be returned.")
        return self._filtered_data

    @property
    def settings(self):
        return self._ranks[0].settings

    def __repr__(self) -> str:
        n_chans = sum(p._n_chans for p in self._ranks)
        return f"MultiDevicePARRM object | devices {self._devices} | Data: ({n_chans} channels x {self._ranks[0]._n_samples} times)"


def filter_host_sharded(filt: np.ndarray, x: np.ndarray, out: np.ndarray | None = None, devices=None,
                        out_dtype=None, chunk_samples: int = 0) -> np.ndarray:
    """BASELINE configs[4] as a package call: a HOST recording ``x[C, N]`` (float32 or float64; page-locked or not)
    filtered with the filter array ``filt`` through every device of this process -- contiguous channel blocks,
    one host thread per device, each block streamed through its device in time chunks with a half-width halo
    (``parrm_filter_host``), no exchange.  Returns the filtered recording (``out`` if given).

    Arrays in page-locked memory (``torch.empty(..., pin_memory=True).numpy()``, ``hipHostMalloc``) are used in
    place -- the 80 GB/s path; anything else is staged through page-locked buffers of the library's own.  The
    library does not lock caller memory itself: a range registered, unregistered and registered again faulted
    under DMA (``profiles/r03_host_register_fault.txt``); ``PARRM_HOST_PIN_PARENTS=1`` restores the round-2
    behaviour of locking large parent arrays once per call."""
    import torch

    _hip.require_gpu()
    if x.ndim != 2:
        raise ValueError("`x` must be a 2D array.")
    if devices is None:
        devices = list(range(torch.cuda.device_count()))
    devices = [int(d) for d in devices][: max(1, min(len(devices), x.shape[0]))]
    out_dtype = np.dtype(out_dtype or (np.float32 if x.dtype == np.float32 else np.float64))
    if out is None:
        out = np.empty(x.shape, dtype=out_dtype)
    if out.shape != x.shape or out.dtype != out_dtype or not out.flags.c_contiguous or not x.flags.c_contiguous:
        raise ValueError("`x` and `out` must be C-contiguous arrays of one shape (`out` of the output dtype).")
    pinned = []
    import os

    for arr in (x, out) if os.environ.get("PARRM_HOST_PIN_PARENTS") else ():
        if arr.nbytes >= (64 << 20):
            try:
                _hip.pin_host(arr)
                pinned.append(arr)
            except _hip.HipLibraryError:
                pass  # lock limit: the blocks are staged instead
    errors = [None] * len(devices)

    def work(r):
        try:
            lo, hi = channel_shard(x.shape[0], r, len(devices))
            if hi > lo:
                torch.cuda.set_device(devices[r])
                plan = _hip.FilterPlan(filt, device=devices[r])
                plan.apply_host(x[lo:hi], out_dtype=out_dtype.type, chunk_samples=chunk_samples, out=out[lo:hi])
        except BaseException as exc:  # noqa: BLE001
            errors[r] = exc

    try:
        threads = [threading.Thread(target=work, args=(r,), name=f"parrm-host{r}") for r in range(len(devices))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        for arr in pinned:
            _hip.unpin_host(arr)
    for e in errors:
        if e is not None:
            raise e
    return out


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
