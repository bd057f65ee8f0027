"""Multi-GPU layout of the PARRM hot path: one process per GPU, no data-path collectives.

What shards:

* ``filter_data``: every channel row is filtered independently (parrm.py:861-866 broadcasts one
  kernel over the columns) -> contiguous channel blocks, one per rank; an N-rank result is the
  concatenation of the rank results, bit-identical to the 1-rank result.
* ``find_period`` candidate grid: candidates are independent (parrm.py:445-454 maps over them)
  -> contiguous slices of the grid, one per rank, when every rank holds the gathered stage matrix.
* independent recordings (per-site period estimation, examples/plot_example_dbs_data.py:52-98):
  one ``PARRM`` per rank -- what ``bench.py`` times (weak scaling).

``torch.distributed`` is used for the benchmark's barrier / MAX-over-ranks timing only; the
helpers here are plain index arithmetic and run anywhere (they are exercised with world_size-2
``gloo`` tests on CPU).
"""

from __future__ import annotations

import time


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
