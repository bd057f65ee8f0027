"""``PARRM`` façade: PyPARRM's class surface over the MI355X engine.

Mirrors ``pyparrm.PARRM`` (reference ``src/pyparrm/parrm.py:18-936``): same constructor and
method signatures, same validation order and messages (they are the behavioural contract the
reference tests pin, ``tests/test_parrm.py:112-330``), same call-order state machine, same
``settings`` layout.  What differs is where the arithmetic runs:

* ``find_period``  -> device: statistics pass, column gather, batched harmonic-regression
  objective (``parrm_absdiff_mean`` / ``parrm_gather_standardise`` / ``parrm_fit_errors``); host:
  index draws and candidate grids with the reference's NumPy expressions, and a Nelder-Mead that
  makes SciPy ``fmin``'s decisions but evaluates its candidates in device batches.
* ``create_filter`` -> host (microseconds; the tap set is a step function of the period, so it is
  built with the same NumPy expressions to stay bit-identical, parrm.py:803-833).
* ``filter_data``  -> device: closed-form phase-neighbour stencil (``parrm_filter_apply``).

There is no CPU fallback: without the HIP library and a GPU, ``find_period`` / ``filter_data``
raise.  Extensions over the reference (additive): ``data`` may be a 2-D CUDA ``torch.Tensor``
(results then stay on the device); ``PARRM.cache_on_device`` (below).

Host recordings are re-read on every ``find_period()`` / ``filter_data()`` call, exactly like the
reference (which reads ``self._data`` each time, parrm.py:274, :861): an in-place edit of the array
between two calls is picked up.  Within one call the device copy is shared by all stages.  Setting
``parrm.cache_on_device = True`` keeps the copy between calls (one upload for
``find_period`` + ``filter_data``; what an interactive explorer wants) -- the caller then promises
not to modify the array in place, or calls :meth:`PARRM.release_device_cache` after doing so.
"""

from __future__ import annotations

import copy
import os
from multiprocessing import cpu_count

import numpy as np

from . import _hip
from ._neldermead import fmin_lockstep_requests

_PERIOD_FAILURE = (
    "The period cannot be estimated from the data. Check that your data "
    "does not contain infs or NaNs."
)
_NO_PERIOD = (
    "The period has not yet been estimated. The `find_period` method must be called first."
)
_DIRECTIONS = ["both", "past", "future"]

# stage schedule of the period search (parrm.py:288-296)
_STAGE_LENGTHS = (5000, 10000, 25000)
_STAGE_IGNORE = (0.0, 0.0, 0.95)
_STAGE_BANDWIDTHS = (5, 10, 20)
_STAGE_LAMBDA = 1.0
# stage grids up to this many candidates join the optimiser steps in find_period_batched's shared call
_SMALL_GRID = 512
_NM_IN_PYTHON = __import__("os").environ.get("PARRM_NM_PYTHON", "0") == "1"  # step through _neldermead.py (A/B, debugging)
# above this size a host recording is streamed through the device instead of cached on it
_DEVICE_CACHE_BYTES = 96 << 30
_HALF_WIDTH_MEMO: dict = {}  # (limit, omit, period, period half-width) -> default filter half-width
_FILTER_MEMO: dict = {}      # (half-width, period, period half-width, omit, direction) -> filter array (copied out)
_PIPELINE_BYTES = 256 << 20        # host recordings from this size on are filtered in overlapped channel blocks
_PIPELINE_BLOCK_BYTES = 256 << 20  # ... of about this size


def _is_number(value) -> bool:
    return isinstance(value, (int, float))


def _is_device_tensor(value) -> bool:
    mod = type(value).__module__
    return mod.startswith("torch") and hasattr(value, "is_cuda") and bool(value.is_cuda)


class PARRM:
    """Remove periodic stimulation artefacts with PARRM (Dastin-van Rijn et al., 2021).

    Call order: :meth:`find_period` -> :meth:`create_filter` -> :meth:`filter_data`.

    Parameters
    ----------
    data : numpy.ndarray, shape of [channels, times]
        Recording to clean.  (Extension: a 2-D CUDA ``torch.Tensor`` is accepted too.)
    sampling_freq : int | float
        Sampling frequency of ``data`` in Hz.
    artefact_freq : int | float
        Stimulation frequency in Hz.
    verbose : bool (default True)
        Print progress messages.
    """

    _data = None
    _standard_data = None
    _filtered_data = None
    _sampling_freq = None
    _artefact_freq = None
    _verbose = None
    _period = None
    _search_span = None  # (first, last, count) of the sorted search samples
    _search_samples_array = None
    _assumed_periods = None
    _outlier_boundary = None
    _random_seed = None
    _n_jobs = None
    _filter = None
    _filter_half_width = None
    _omit_n_samples = None
    _filter_direction = None
    _period_half_width = None

    #: keep the device copy of a host recording between calls (see the module docstring)
    cache_on_device = False
    #: a filter whose generated kernel has to be compiled first (~1.7 s of hipRTC for a tap geometry nobody has built
    #: yet) is filtered by the generic kernels -- same results within the parity bar -- while a worker thread builds
    #: the generated one: no ``filter_data`` call waits for a compiler (the reference's parameter explorer re-filters
    #: on every widget event, _utils/_plotting.py:568-584).  False: the first large call compiles, as in round 3.
    background_compile = True
    #: NumPy in -> NumPy out ``filter_data`` of a large host recording reads its result back into page-locked memory up
    #: to this many bytes (the pinned allocator keeps such blocks for the life of the process, rounded up to a power of
    #: two); larger results, or 0, use an ordinary array (slower read-back, no lasting footprint).
    pinned_result_max_bytes = 8 << 30

    # device-side state: rebuilt lazily, never copied or pickled (``__deepcopy__`` / ``__getstate__``)
    _DEVICE_STATE = ("_d_data", "_d_data_src", "_d_scale", "_plans", "_last_plan", "_in_call")

    # ------------------------------------------------------------------ construction (a1)
    def __init__(self, data, sampling_freq, artefact_freq, verbose=True) -> None:
        self._check_init_inputs(data, sampling_freq, artefact_freq, verbose)
        self._n_chans, self._n_samples = (int(s) for s in self._data.shape)
        self._reset_device_state()
        self._trace = None

    def _reset_device_state(self) -> None:
        self._d_data = None  # device copy of a host recording (lazy)
        self._d_data_src = None  # the host array object that copy was made from
        self._d_scale = None
        self._plans = {}  # device index -> FilterPlan of the current filter
        self._last_plan = None
        self._in_call = False  # inside find_period()/filter_data(): the device copy is current

    # ------------------------------------------------------------------ copying (explorer: _plotting.py:115)
    def __deepcopy__(self, memo):
        """``copy.deepcopy(parrm)`` -- the first thing the reference's explorer does
        (_utils/_plotting.py:115).  Host state is copied like the reference's plain object would be;
        device state (cached recording, filter plans: raw device pointers) is dropped and rebuilt
        lazily by the copy; a CUDA-tensor recording is shared, not cloned (it is never written)."""
        clone = object.__new__(type(self))
        memo[id(self)] = clone
        for name, value in self.__dict__.items():
            if name in self._DEVICE_STATE:
                continue
            if _is_device_tensor(value):
                setattr(clone, name, value)
            else:
                setattr(clone, name, copy.deepcopy(value, memo))
        clone._reset_device_state()
        return clone

    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in self._DEVICE_STATE}
        if _is_device_tensor(state.get("_data")):
            raise TypeError("a PARRM object holding a CUDA tensor cannot be pickled; pass a NumPy array")
        if _is_device_tensor(state.get("_filtered_data")):
            state["_filtered_data"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._reset_device_state()

    @property
    def _plan(self):
        """The filter plan most recently used by :meth:`filter_data` (``None`` before that)."""
        return self._last_plan

    def _check_init_inputs(self, data, sampling_freq, artefact_freq, verbose) -> None:
        """parrm.py:112-140 (first failing check wins)."""
        if not isinstance(data, np.ndarray) and not _is_device_tensor(data):
            raise TypeError("`data` must be a NumPy array.")
        if data.ndim != 2:
            raise ValueError("`data` must be a 2D array.")
        self._data = data  # held by reference, never mutated (parrm.py:124)
        for name, value in (("sampling_freq", sampling_freq), ("artefact_freq", artefact_freq)):
            if not _is_number(value):
                raise TypeError(f"`{name}` must be an int or a float.")
            if value <= 0:
                raise ValueError(f"`{name}` must be > 0.")
            setattr(self, f"_{name}", value)
        if not isinstance(verbose, bool):
            raise TypeError("`verbose` must be a bool.")
        self._verbose = verbose

    @property
    def _search_samples(self):
        """Sorted sample indices the period search may use (the reference's attribute of the same
        name); the default ``arange(N-1)`` is materialised on first access."""
        if self._search_samples_array is None and self._search_span is not None:
            self._search_samples_array = np.arange(self._search_span[2])
        return self._search_samples_array

    @_search_samples.setter
    def _search_samples(self, value) -> None:
        self._search_samples_array = value
        self._search_span = None if value is None else (value[0], value[-1], value.shape[0])

    def __repr__(self) -> str:
        return (
            f"PARRM object | Data: ({self._n_chans} channels x "
            f"{self._n_samples} times) | Period: {self._period:.4f}"
        )

    def _say(self, text: str) -> None:
        if self._verbose:
            print(text)

    # ------------------------------------------------------------------ device residency
    def _device_recording(self, data=None):
        """Float device tensor [C, N] for ``data`` (default: the object's own recording)."""
        torch = _hip.require_gpu()
        own = data is None or data is self._data
        src = self._data if own else data
        if _is_device_tensor(src):
            t = src
            if t.dtype not in (torch.float32, torch.float64):
                t = t.to(torch.float64)
            return t if t.stride(-1) == 1 else t.contiguous()
        # The device copy belongs to one host array OBJECT (the reference's explorer rebinds `_data`
        # to a time slice, _plotting.py:140-141) and, unless the caller opted into caching, to the
        # find_period()/filter_data() call that made it: the reference re-reads `self._data` on every
        # call, so an in-place edit between calls must be seen here too.
        if own and self._d_data is not None and self._d_data_src is src and (self._in_call or self.cache_on_device):
            return self._d_data
        host = src
        if host.dtype not in (np.float32, np.float64):
            host = host.astype(np.float64)  # the reference promotes through float64 arithmetic
        t = torch.from_numpy(np.ascontiguousarray(host)).cuda()
        if own:
            self._d_data, self._d_data_src = t, src
            self._d_scale = None
            self._in_call = True  # cleared by _end_call()
        return t

    def _end_call(self) -> None:
        """End of a public call: without ``cache_on_device`` the device copy of a host recording
        is not trusted (nor kept) beyond it."""
        self._in_call = False
        if not self.cache_on_device:
            self._d_data = self._d_data_src = self._d_scale = None

    def release_device_cache(self) -> None:
        """Drop the cached device copy of a host recording (with ``cache_on_device``: call this
        after modifying ``data`` in place)."""
        self._d_data = None
        self._d_data_src = None
        self._d_scale = None
        self._in_call = False

    # ------------------------------------------------------------------ find_period (a2-a8)
    def find_period(
        self,
        search_samples=None,
        assumed_periods=None,
        outlier_boundary=3.0,
        random_seed=None,
        n_jobs=1,
    ) -> None:
        """Estimate the artefact period (reference: parrm.py:148-194).

        ``n_jobs`` is validated as in the reference but not used: candidate periods are
        evaluated as one device batch instead of a thread-pool map.
        """
        self._say("\nFinding the artefact period...")
        self._reset_result_attrs()
        self._check_sort_find_stim_period_inputs(
            search_samples, assumed_periods, outlier_boundary, random_seed, n_jobs
        )
        self._in_call = False  # a new search always starts from the host array as it is now
        if not self.cache_on_device:
            self._d_data = self._d_data_src = None
        try:
            self._standardise_data()
            self._optimise_period_estimate()
        finally:
            self._end_call()
        self._say("    ... Artefact period found\n")

    def _reset_result_attrs(self) -> None:
        """parrm.py:196-211: a new search invalidates everything downstream (not ``_n_jobs``)."""
        for name in (
            "_standard_data", "_filtered_data", "_period", "_search_samples", "_assumed_periods",
            "_outlier_boundary", "_random_seed", "_filter", "_filter_half_width",
            "_omit_n_samples", "_filter_direction", "_period_half_width",
        ):
            setattr(self, name, None)
        self._plans = {}
        self._last_plan = None
        self._d_scale = None

    def _check_sort_find_stim_period_inputs(
        self, search_samples, assumed_periods, outlier_boundary, random_seed, n_jobs
    ) -> None:
        """parrm.py:213-270."""
        if search_samples is not None and not isinstance(search_samples, np.ndarray):
            raise TypeError("`search_samples` must be a NumPy array or None.")
        if search_samples is None:
            # the reference materialises and sorts arange(N-1) here (parrm.py:225-228) but only ever
            # reads its first entry, last entry and length (:290,:352,:360,:364); at 10 M samples
            # that costs more than the whole device search, so the array is built on first access
            # of `_search_samples` / `settings` instead (same values)
            self._search_span = (0, self._n_samples - 2, self._n_samples - 1)
            self._search_samples_array = None
        else:
            if search_samples.ndim != 1:
                raise ValueError("`search_samples` must be a 1D array.")
            search_samples = np.sort(search_samples)
            if search_samples[0] < 0 or search_samples[-1] >= self._n_samples:
                raise ValueError("Entries of `search_samples` must lie in the range [0, n_samples).")
            self._search_samples = search_samples

        if assumed_periods is not None and not isinstance(assumed_periods, (int, float, tuple)):
            raise TypeError("`assumed_periods` must be an int, a float, a tuple, or None.")
        if assumed_periods is None:
            assumed_periods = (self._sampling_freq / self._artefact_freq,)
        elif _is_number(assumed_periods):
            assumed_periods = (assumed_periods,)
        elif not all(_is_number(entry) for entry in assumed_periods):
            raise TypeError("If a tuple, entries of `assumed_periods` must be ints or floats.")
        self._assumed_periods = assumed_periods

        if not _is_number(outlier_boundary):
            raise TypeError("`outlier_boundary` must be an int or a float.")
        if outlier_boundary <= 0:
            raise ValueError("`outlier_boundary` must be > 0.")
        self._outlier_boundary = outlier_boundary

        if random_seed is not None and not isinstance(random_seed, int):
            raise TypeError("`random_seed` must be an int or None.")
        if random_seed is not None:
            self._random_seed = random_seed

        self._n_jobs = _checked_n_jobs(n_jobs)

    def _streams_from_host(self) -> bool:
        """True when the recording stays on the host (a memory-mapped file, or an array larger than
        the device cache budget): ``find_period`` then streams its one full pass in time chunks and
        gathers the <= 25 001 stage columns on the host, ``filter_data`` streams as well."""
        data = self._data
        return isinstance(data, np.ndarray) and (isinstance(data, np.memmap) or data.nbytes > _DEVICE_CACHE_BYTES)

    def _standardise_data(self) -> None:
        """Device statistics pass for parrm.py:272-280.

        Only ``scale[c] = mean|diff|`` is produced (one HBM read of the recording); the
        standardised array itself is never materialised -- each stage gathers and scales the
        <= 25 001 columns it consumes (``_stage_matrix``)."""
        if self._n_samples < 2:
            raise ValueError("`data` must have at least 2 samples to estimate a period.")
        kind = getattr(self._data, "dtype", None)
        if isinstance(self._data, np.ndarray) and kind is not None and not np.issubdtype(kind, np.floating):
            # the reference's in-place divide raises for integer recordings (parrm.py:275)
            raise TypeError("`data` must have a floating-point dtype to estimate the period.")
        if self._streams_from_host():
            self._d_scale = self._streamed_absdiff_mean()
            return
        self._d_scale = _hip.absdiff_mean(self._device_recording())

    def _streamed_absdiff_mean(self, chunk_bytes: int = 1 << 30):
        """``mean|diff|`` per channel of a host-resident recording, one time chunk on the device at a
        time: chunk k contributes ``mean_k * n_k`` (consecutive chunks share one sample, so every
        difference is counted once); the chunk sums are added in float64 in chunk order."""
        torch = _hip.require_gpu()
        data = self._data
        step = max(int(chunk_bytes // (data.dtype.itemsize * self._n_chans)), 2)
        total = None
        for lo in range(0, self._n_samples - 1, step):
            hi = min(lo + step, self._n_samples - 1)  # differences lo .. hi-1 need samples lo .. hi
            host = np.array(data[:, lo:hi + 1])  # a fresh, writable block (the source may be a read-only map)
            if host.dtype not in (np.float32, np.float64):
                host = host.astype(np.float64)
            part = _hip.absdiff_mean(torch.from_numpy(host).cuda()) * float(hi - lo)
            total = part if total is None else total + part
        return total / float(self._n_samples - 1)

    def _stage_matrix(self, indices: np.ndarray):
        torch = _hip.require_gpu()
        if indices.shape[0] == 0 or indices[0] < 0 or indices[-1] >= self._n_samples - 1:
            raise IndexError("period-search sample indices fall outside the differenced recording")
        if self._streams_from_host():
            # gather the two samples of every difference on the host and let the device kernel do the
            # same arithmetic on the compact array: x'[c, 2j] = x[c, idx_j], x'[c, 2j+1] = x[c, idx_j + 1]
            # (float32 and float64 keep their type -- the reference differences a float32 recording in float32,
            # parrm.py:272-280 -- anything else is promoted, as on the resident path and in the statistics pass)
            pair_dtype = self._data.dtype if self._data.dtype in (np.float32, np.float64) else np.float64
            pairs = np.empty((self._n_chans, 2 * indices.shape[0]), dtype=pair_dtype)
            pairs[:, 0::2] = self._data[:, indices]
            pairs[:, 1::2] = self._data[:, indices + 1]
            x = torch.from_numpy(pairs).cuda()
            compact = torch.arange(0, 2 * indices.shape[0], 2, dtype=torch.int64, device=x.device)
            y = _hip.gather_standardise(x, compact, self._d_scale, self._outlier_boundary)
            d_idx = _hip.upload_indices(indices, x.device)
            return y, d_idx
        x = self._device_recording()
        d_idx = _hip.upload_indices(indices, x.device)
        y = _hip.gather_standardise(x, d_idx, self._d_scale, self._outlier_boundary)
        return y, d_idx

    def _grid_errors(self, y, d_idx, grid: np.ndarray, bandwidth: int, workspace) -> np.ndarray:
        """Errors of a whole candidate grid (the ordered map of parrm.py:445-454).  One device call
        here; ``sharding.ShardedPARRM`` cuts the grid into per-rank slices instead."""
        return _hip.fit_errors(y, d_idx, grid, bandwidth, _STAGE_LAMBDA, workspace)

    def _search_requests(self):
        """The period search (parrm.py:282-325: three coarse-to-fine stages + an un-regularised
        polish) as a generator of objective evaluations: yields :class:`FitRequest` objects, expects
        the float64 error vector of each to be sent back, and leaves ``_period`` / ``_trace`` set when
        it finishes.  ``find_period`` answers the requests one by one; :func:`find_period_batched`
        advances several searches together and answers all their small requests in one device call."""
        rng = np.random.default_rng(self._random_seed)
        estimate = self._assumed_periods
        lengths = np.unique(
            [int(np.min((self._search_span[2], n))) for n in _STAGE_LENGTHS]
        )
        trace = []
        stage = None
        # Every stage's sample indices and stage matrix are made NOW: they depend on the recording, its scale and
        # the seed only (the reference draws stage 3's indices from the same generator at the same point of its
        # sequence, parrm.py:284, :369-371), so the three gathers queue on the device right behind the statistics
        # pass -- which the host would otherwise sit out -- instead of each waiting for the previous stage's
        # optimiser while the device idles through 25 000 PCG64 draws and a sort.
        plans = []
        for run, (use_n, ignore, bandwidth) in enumerate(
            zip(lengths, _STAGE_IGNORE, _STAGE_BANDWIDTHS), start=1
        ):
            indices = self._get_centre_indices(use_n, ignore, rng)
            plans.append((run, indices, int(np.min((bandwidth, indices.shape[0] // 4)))))
        # (so is the first stage's candidate grid -- it depends on the assumed periods only --: built before the index
        # upload below blocks on the statistics pass, not in the gap between the gathers and the first grid's kernels)
        first_grid = self._get_possible_periods(estimate, 1)
        matrices = [self._stage_matrix(indices) for _, indices, _ in plans]
        for (run, indices, bandwidth), (y, d_idx) in zip(plans, matrices):
            stage = (y, d_idx)
            grid = first_grid if run == 1 else self._get_possible_periods(estimate, run)
            errors = yield FitRequest(y, d_idx, grid, bandwidth, _STAGE_LAMBDA, True)
            ranked, ranked_errors = _rank_candidates(grid, errors)
            evals = []
            estimate = yield from _refine_candidates(ranked, ranked_errors, y, d_idx, bandwidth, evals)
            trace.append(
                {"indices": indices, "bandwidth": bandwidth, "grid": grid, "errors": errors,
                 "estimate": float(estimate[0]), "refine_evals": evals}
            )
        if not np.isfinite(estimate[0]):
            raise ValueError(_PERIOD_FAILURE)

        # final polish: lambda = 0, bandwidth 20 *unclipped*, last stage's indices (parrm.py:524-550)
        y, d_idx = stage
        final_bw = _STAGE_BANDWIDTHS[-1]
        final_evals = []
        result = yield from _fmin_requests([estimate[0]], y, d_idx, final_bw, 0.0, final_evals)
        self._period = result[0][0][0]
        trace.append({"final_bandwidth": final_bw, "final_evals": final_evals})
        self._trace = trace

    def _answer(self, request, workspace) -> np.ndarray:
        """Evaluate one request of this object's own search."""
        if request.is_grid:
            return self._grid_errors(request.y, request.d_idx, request.periods, request.bandwidth, workspace)
        if isinstance(request, NelderMeadRequest):
            # the whole refinement inside the library: no Python between two optimiser batches (PARRM_NM_PYTHON=1
            # steps through pyparrm_amd/_neldermead.py instead: same batches, same decisions)
            if _NM_IN_PYTHON:
                return None
            return _hip.NativeNelderMead(request.starts).minimise_fit(request.y, request.d_idx, request.bandwidth,
                                                                     request.lambda_, workspace)
        return _hip.fit_errors(request.y, request.d_idx, request.periods, request.bandwidth, request.lambda_, workspace)

    def _optimise_period_estimate(self) -> None:
        """Drive :meth:`_search_requests` to the end, one device call per request."""
        ws = _hip.FitWorkspace()
        requests = self._search_requests()
        try:
            request = next(requests)
            while True:
                request = requests.send(self._answer(request, ws))
        except StopIteration:
            pass

    def _get_centre_indices(self, use_n_samples, ignore_portion, random_state) -> np.ndarray:
        """Sample indices for one stage (parrm.py:327-374): a contiguous centre block, or -- when
        the block would not exceed ``ignore_portion`` of the recording -- unique random draws
        from the central part.  Kept on the host so the PCG64 draws are the reference's."""
        first, last = self._search_span[0], self._search_span[1]
        centre2 = first + last
        lo = int(np.ceil((centre2 - use_n_samples) / 2))
        hi = int(np.floor((centre2 + use_n_samples) / 2))
        if self._n_samples * ignore_portion < hi - lo:
            return np.arange(lo, hi + 1)
        margin = (1.0 - ignore_portion) / 2.0 * self._n_samples
        lo = int(first + np.floor(margin))
        hi = int(last - np.ceil(margin))
        draws = random_state.integers(0, hi - lo, np.min((use_n_samples, hi - lo)))
        return np.unique(draws) + lo

    @staticmethod
    def _get_possible_periods(estimated_period, run: int) -> np.ndarray:
        """Candidate grid (parrm.py:376-405): +-1 % in 1e-4 steps and +-0.1 % in 1e-5 steps
        around every estimate, both shrunk by ``run``; 1-ulp near-duplicates survive ``unique``."""
        rel = np.concatenate(
            (
                1 + np.arange(-1e-2, 1e-2 + 1e-4, 1e-4) / run,
                1 + np.arange(-1e-3, 1e-3 + 1e-5, 1e-5) / run,
            )
        )
        # (every estimate times every factor: the products of the reference's loop, as one outer product)
        estimates = np.asarray(estimated_period, dtype=np.float64).reshape(-1)
        return np.unique(np.multiply.outer(estimates, rel))

    # ------------------------------------------------------------------ explorer (out of scope)
    def explore_filter_params(self, time_range=None, time_res=0.01, freq_range=None, freq_res=5.0,
                              n_jobs=1) -> None:
        """Interactive parameter explorer (parrm.py:634-687): GUI, outside this engine's scope
        (SURVEY.md section 8f-1).  The call-order check and the argument contract are kept."""
        self._say("Opening the filter parameter explorer...")
        if self._period is None:
            raise ValueError(_NO_PERIOD)
        self._check_explorer_inputs(time_range, time_res, freq_range, freq_res, n_jobs)
        raise NotImplementedError(
            "explore_filter_params is a matplotlib GUI of the reference and is not part of the "
            "MI355X hot path; create_filter()/filter_data() can be called repeatedly instead."
        )

    def _check_explorer_inputs(self, time_range, time_res, freq_range, freq_res, n_jobs) -> None:
        """Argument contract of the explorer (reference: _utils/_plotting.py:101-186), kept so that
        callers get the reference's errors in the reference's order before the GUI is declined."""
        max_time = self._n_samples / self._sampling_freq
        nyquist = self._sampling_freq / 2

        def check_range(value, name, default, bad_bounds, bounds_text):
            value = default if value is None else value
            if not isinstance(value, list) or not all(_is_number(entry) for entry in value):
                raise TypeError(f"`{name}` must be a list of ints or floats.")
            if len(value) != 2:
                raise ValueError(f"`{name}` must have a length of 2.")
            if bad_bounds(value):
                raise ValueError(f"Entries of `{name}` must lie in the range {bounds_text}.")
            if value[0] >= value[1]:
                raise ValueError(f"`{name}[1]` must be > `{name}[0]`.")
            return value

        time_range = check_range(time_range, "time_range", [0, max_time],
                                 lambda v: v[0] < 0 or v[1] > max_time, "[0, max. time]")
        last_sample = np.arange(time_range[0] * self._sampling_freq,
                                time_range[1] * self._sampling_freq).astype(int)[-1]
        if not _is_number(time_res):
            raise TypeError("`time_res` must be an int or a float.")
        if time_res <= 0 or time_res >= last_sample / self._sampling_freq:
            raise ValueError("`time_res` must lie in the range (0, max. time).")
        check_range(freq_range, "freq_range", [1, nyquist],
                    lambda v: v[0] <= 0 or v[1] > nyquist, "(0, Nyquist frequency]")
        if not _is_number(freq_res):
            raise TypeError("`freq_res` must be an int or a float.")
        if freq_res <= 0 or freq_res > nyquist:
            raise ValueError("`freq_res` must lie in the range (0, Nyquist frequency].")
        _checked_n_jobs(n_jobs)

    # ------------------------------------------------------------------ create_filter (a9, a10)
    def create_filter(self, filter_half_width=None, omit_n_samples=0, filter_direction="both",
                      period_half_width=None) -> None:
        """Design the PARRM filter (reference: parrm.py:689-737)."""
        self._say("Creating the filter...")
        if self._period is None:
            raise ValueError(_NO_PERIOD)
        self._check_sort_create_filter_inputs(
            filter_half_width, omit_n_samples, filter_direction, period_half_width
        )
        self._generate_filter()
        self._say("    ... Filter created\n")

    def _check_sort_create_filter_inputs(self, filter_half_width, omit_n_samples, filter_direction,
                                         period_half_width) -> None:
        """parrm.py:739-786; order: omit -> period half-width -> half-width -> direction."""
        half_max = (self._n_samples - 1) // 2
        if not isinstance(omit_n_samples, int):
            raise TypeError("`omit_n_samples` must be an int.")
        if omit_n_samples < 0 or omit_n_samples >= half_max:
            raise ValueError("`omit_n_samples` must lie in the range [0, (no. of samples - 1) // 2).")
        self._omit_n_samples = omit_n_samples

        if period_half_width is None:
            period_half_width = self._period / 50
        if not _is_number(period_half_width):
            raise TypeError("`period_half_width` must be an int or a float.")
        if period_half_width <= 0 or period_half_width > self._period:
            raise ValueError("`period_half_width` must be lie in the range (0, period].")
        self._period_half_width = period_half_width

        if filter_half_width is None:  # needs the two settings above
            filter_half_width = self._get_filter_half_width()
        if not isinstance(filter_half_width, int):
            raise TypeError("`filter_half_width` must be an int.")
        if filter_half_width <= omit_n_samples or filter_half_width > half_max:
            raise ValueError(
                "`filter_half_width` must lie in the range (`omit_n_samples`, "
                "(no. of samples - 1) // 2]."
            )
        self._filter_half_width = filter_half_width

        if not isinstance(filter_direction, str):
            raise TypeError("`filter_direction` must be a str.")
        if filter_direction not in _DIRECTIONS:
            raise ValueError(f"`filter_direction` must be one of {_DIRECTIONS}.")
        self._filter_direction = filter_direction

    def _get_filter_half_width(self) -> int:
        """Smallest half-width covering 50 in-phase offsets beyond the omitted centre, capped at
        (N-1)//2 (parrm.py:788-801; the `>= period + half-width` clause there never fires).  A pure function of
        (limit, omitted samples, period, period half-width): remembered (the reference's explorer, and any loop
        over recordings of one study, asks the same question again and again)."""
        key = ((self._n_samples - 1) // 2, self._omit_n_samples, float(self._period), float(self._period_half_width))
        hit = _HALF_WIDTH_MEMO.get(key)
        if hit is None:
            if len(_HALF_WIDTH_MEMO) > 256:
                _HALF_WIDTH_MEMO.clear()
            hit = _HALF_WIDTH_MEMO[key] = self._walk_filter_half_width()
        return hit

    def _walk_filter_half_width(self) -> int:
        limit = (self._n_samples - 1) // 2
        width, hits = self._omit_n_samples, 0
        # the reference walks one offset at a time; the same walk in vectorised chunks
        chunk = 4096
        while hits < 50 and width < limit:
            widths = np.arange(width + 1, min(limit, width + chunk) + 1)
            phase = np.mod(widths, self._period)
            hit = (phase <= self._period_half_width) | (phase >= self._period + self._period_half_width)
            seen = hits + np.cumsum(hit)
            done = np.flatnonzero(seen >= 50)
            if done.size:
                return int(widths[done[0]])
            hits, width = int(seen[-1]), int(widths[-1])
            chunk *= 4
        return width

    def _generate_filter(self) -> None:
        """Dense filter array (parrm.py:803-833): 1 at the centre, -1/S on every offset whose
        phase is within ``period_half_width`` of the centre's, 0 elsewhere."""
        hw, period, phw = self._filter_half_width, self._period, self._period_half_width
        key = (hw, float(period), float(phw), self._omit_n_samples, self._filter_direction)
        master = _FILTER_MEMO.get(key)
        if master is not None:  # (same design as before: a fresh copy -- `filter` hands out the internal array)
            self._filter = master.copy()
            self._plans = {}
            self._last_plan = None
            return
        offsets = np.arange(-hw, hw + 1)
        phase = np.mod(offsets, period)
        chosen = ((phase <= phw) | (phase >= period - phw)) & (np.abs(offsets) > self._omit_n_samples)
        if self._filter_direction == "past":
            chosen &= offsets <= 0
        elif self._filter_direction == "future":
            chosen &= offsets > 0
        n_taps = int(np.count_nonzero(chosen))
        if n_taps == 0:
            raise RuntimeError(
                "A suitable filter cannot be created with the specified settings. Try "
                "reducing the number of omitted samples and/or increasing the filter "
                "half-width."
            )
        taps = np.zeros(offsets.shape, dtype=np.float64)
        taps[chosen] = 1.0
        taps = -taps / np.max((taps.sum(), np.finfo(np.float64).eps))
        taps[hw] = 1
        if len(_FILTER_MEMO) > 64:
            _FILTER_MEMO.clear()
        _FILTER_MEMO[key] = taps.copy()
        self._filter = taps
        self._plans = {}  # device tables are rebuilt lazily for the new taps
        self._last_plan = None

    # ------------------------------------------------------------------ filter_data (a11)
    def filter_data(self, data=None):
        """Apply the filter (reference: parrm.py:835-875) and return the cleaned recording.

        float64 out for any input dtype (the reference's rule).  NumPy in -> NumPy out; a CUDA
        tensor in -> CUDA tensor out.  The result is also kept as ``filtered_data``.
        """
        self._say("Filtering the data...")
        if self._filter is None:
            raise ValueError(
                "The filter has not yet been created. The `create_filter` method must "
                "be called first."
            )
        data = self._check_sort_filter_data_inputs(data)
        try:
            if isinstance(data, np.ndarray) and (isinstance(data, np.memmap) or data.nbytes > _DEVICE_CACHE_BYTES):
                filtered = self._plan_for(None).apply_host(np.ascontiguousarray(data))
            elif (isinstance(data, np.ndarray) and data.nbytes >= _PIPELINE_BYTES and data.shape[0] >= 2
                  and type(self)._total_chans is PARRM._total_chans and not (self.cache_on_device and data is self._data)):
                filtered = self._filter_host_pipelined(data)
            else:
                x = self._device_recording(data)
                y = self._plan_for(x.device).apply(x, total_chans=self._total_chans(x))
                filtered = _hip.to_host_numpy(y) if isinstance(data, np.ndarray) else y
        finally:
            self._end_call()
        self._filtered_data = filtered
        self._say("    ... Data filtered\n")
        return self._filtered_data

    def _filter_host_pipelined(self, data: np.ndarray) -> np.ndarray:
        """NumPy in -> NumPy out for a large host recording (the reference's calling convention, parrm.py:835-875:
        ndarray in, fresh ndarray out): contiguous channel blocks travel host -> device -> kernel -> page-locked
        result on two alternating streams, so the upload of block k + 1 (a pageable copy, driven by this thread)
        overlaps the kernel and the read-back of block k.  Channels are independent (:861-866) and every block is
        cut like the whole recording (``total_chans``), so the result equals the one-piece call bit for bit."""
        torch = _hip.require_gpu()
        n_chans, n_samples = data.shape
        plan = self._plan_for(None)
        dev = torch.device("cuda", plan.device)
        n_blocks = max(2, min(n_chans, -(-data.nbytes // _PIPELINE_BLOCK_BYTES)))
        base, extra = divmod(n_chans, n_blocks)
        bounds, lo = [], 0
        for b in range(n_blocks):
            hi = lo + base + (1 if b < extra else 0)
            bounds.append((lo, hi))
            lo = hi
        rows_max = bounds[0][1] - bounds[0][0]
        in_dtype = torch.float32 if data.dtype == np.float32 else torch.float64
        # The result is read back into page-locked memory (57 GB/s against 10-20 pageable) when it is small enough to
        # afford: torch's pinned allocator rounds a request up to a power of two and never returns a block to the OS, so
        # every call of this size keeps that much of the host page-locked for the life of the process.
        # Beyond `pinned_result_max_bytes` (default 8 GiB; 0 = never pin) the result is an ordinary array.
        out = None
        if 0 < n_chans * n_samples * 8 <= self.pinned_result_max_bytes:
            try:
                out = torch.empty((n_chans, n_samples), dtype=torch.float64, pin_memory=True)
            except RuntimeError:  # the host's page-lock limit
                out = None
        if out is None:
            out = torch.empty((n_chans, n_samples), dtype=torch.float64)
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        x_d = [torch.empty((rows_max, n_samples), dtype=in_dtype, device=dev) for _ in range(2)]
        y_d = [torch.empty((rows_max, n_samples), dtype=torch.float64, device=dev) for _ in range(2)]
        for st in streams:  # the buffers above were allocated on the current stream
            st.wait_stream(torch.cuda.current_stream(dev))
        for k, (lo, hi) in enumerate(bounds):
            rows, b = hi - lo, k & 1
            block = data[lo:hi]
            if block.dtype not in (np.float32, np.float64):
                block = block.astype(np.float64)
            with torch.cuda.stream(streams[b]):
                x_d[b][:rows].copy_(torch.from_numpy(np.ascontiguousarray(block)), non_blocking=True)
                plan.apply(x_d[b][:rows], out=y_d[b][:rows], total_chans=n_chans)
                out[lo:hi].copy_(y_d[b][:rows], non_blocking=True)
        for st in streams:
            st.synchronize()
        return out.numpy()

    def online(self, n_chans=None, dtype=None, out_dtype=np.float64, device=None):
        """Stateful block-by-block filtering with the current filter (``streaming.OnlineFilter``):
        push consecutive blocks, receive every output that has become computable.  Zero latency for
        the one-sided filter that only reaches earlier samples (``filter_direction="future"``, the
        taps ``w > 0`` of parrm.py:819-820)."""
        if self._filter is None:
            raise ValueError(
                "The filter has not yet been created. The `create_filter` method must "
                "be called first."
            )
        from .streaming import OnlineFilter

        if dtype is None:
            dtype = np.float32 if str(getattr(self._data, "dtype", "")).endswith("float32") else np.float64
        return OnlineFilter(self._filter, self._n_chans if n_chans is None else n_chans, dtype=dtype,
                            out_dtype=out_dtype, device=device)

    def _total_chans(self, x):
        """Channels of the whole recording ``x`` is a block of (``sharding.ShardedPARRM`` overrides)."""
        return int(x.shape[0])

    def _plan_for(self, device):
        """Filter plan on ``device`` (default: the current one).  The plan's tables live in that
        device's memory, so plans are kept per device index and built with that device current."""
        torch = _hip.require_gpu()
        index = torch.cuda.current_device() if device is None or device.index is None else device.index
        plan = self._plans.get(index)
        if plan is None:
            plan = self._plans[index] = _hip.shared_filter_plan(self._filter, index)
        # (per call: plans are shared between objects with the same filter, and a channel-sharded object must not have
        # its kernel swapped in mid-run)
        plan.set_background_compile(self.background_compile and type(self)._total_chans is PARRM._total_chans
                                    and os.environ.get("PARRM_COMB_ASYNC", "1") != "0")
        self._last_plan = plan
        return plan

    def _check_sort_filter_data_inputs(self, data):
        """parrm.py:877-886."""
        if data is None:
            data = self._data
        if not isinstance(data, np.ndarray) and not _is_device_tensor(data):
            raise TypeError("`data` must be a NumPy array.")
        if data.ndim != 2:
            raise ValueError("`data` must be a 2D array.")
        return data

    # ------------------------------------------------------------------ properties (a12)
    @property
    def data(self):
        """The original recording."""
        return self._data

    @property
    def period(self):
        """The estimated artefact period, in samples."""
        if self._period is None:
            raise AttributeError("No period has been computed yet.")
        return self._period

    @property
    def filter(self) -> np.ndarray:
        """The PARRM filter (the internal array, not a copy)."""
        if self._filter is None:
            raise AttributeError("No filter has been computed yet.")
        return self._filter

    @property
    def filtered_data(self):
        """The most recently filtered recording."""
        if self._filtered_data is None:
            raise AttributeError("No data has been filtered yet.")
        return self._filtered_data

    @property
    def settings(self) -> dict:
        """Settings behind the current filter (layout of parrm.py:914-936)."""
        if self._period is None or self._filter is None:
            raise AttributeError("Analysis settings have not been established yet.")
        return {
            "data": {"sampling_freq": self._sampling_freq, "artefact_freq": self._artefact_freq},
            "period": {
                "search_samples": self._search_samples,
                "assumed_periods": self._assumed_periods,
                "outlier_boundary": self._outlier_boundary,
                "random_seed": self._random_seed,
            },
            "filter": {
                "filter_half_width": self._filter_half_width,
                "omit_n_samples": self._omit_n_samples,
                "filter_direction": self._filter_direction,
                "period_half_width": self._period_half_width,
            },
        }


def _checked_n_jobs(n_jobs) -> int:
    """parrm.py:262-270."""
    if not isinstance(n_jobs, int):
        raise TypeError("`n_jobs` must be an int.")
    if n_jobs > cpu_count():
        raise ValueError("`n_jobs` must be <= the number of available CPUs.")
    if n_jobs <= 0 and n_jobs != -1:
        raise ValueError("If `n_jobs` is <= 0, it must be -1.")
    return cpu_count() if n_jobs == -1 else n_jobs


_RANK_KEEP = 6  # the <= 5 starts of the refinement (parrm.py:499) + the best candidate that is not refined


def _rank_candidates(periods: np.ndarray, errors: np.ndarray):
    """Sort candidates by error and drop the non-finite ones (parrm.py:456-465).

    What the stage goes on to use of the ranking is its head: the best <= 5 candidates start the refinement and the
    estimate is the arg-min over [their refined errors, every other error] (:499-522), i.e. over the head and the
    best candidate behind it.  A large all-finite grid (the 10 044-candidate stage of BASELINE configs[2]: a full
    ``argsort`` costs the host ~0.2 ms while the device idles) is therefore ranked by selection: the 7 smallest
    errors, in order.  Only where those are strictly increasing is the head of ANY correct ``argsort`` the same six
    candidates in the same order -- NumPy's default sort is not stable and the grids hold exact ties (1-ulp
    neighbours among the periods); otherwise the full sort runs, as the reference's does."""
    n = errors.shape[0]
    if n > 64 and np.isfinite(errors).all():
        head = np.argpartition(errors, _RANK_KEEP)[: _RANK_KEEP + 1]
        head = head[np.argsort(errors[head], kind="stable")]
        if np.all(np.diff(errors[head]) > 0):
            return periods[head[:_RANK_KEEP]], errors[head[:_RANK_KEEP]]
    order = errors.argsort()
    errors = errors[order]
    periods = periods[order[np.isfinite(errors)]]
    if periods.shape == (0,):
        raise ValueError(_PERIOD_FAILURE)
    return periods, errors


class FitRequest:
    """One objective evaluation of a period search: the candidates ``periods`` on the stage matrix
    ``y`` / sample indices ``d_idx`` (device tensors) with harmonics up to ``bandwidth`` and
    regulariser weight ``lambda_``.  ``is_grid`` marks a whole stage grid (hundreds to thousands of
    candidates: throughput-bound) as opposed to one optimiser step (<= 25: latency-bound)."""

    __slots__ = ("y", "d_idx", "periods", "bandwidth", "lambda_", "is_grid")

    def __init__(self, y, d_idx, periods, bandwidth, lambda_, is_grid):
        self.y, self.d_idx, self.periods = y, d_idx, periods
        self.bandwidth, self.lambda_, self.is_grid = bandwidth, lambda_, is_grid


class NelderMeadRequest:
    """A whole lock-step Nelder-Mead refinement (parrm.py:510-517, :545-550) offered to the driver in one piece: a
    driver that can run it natively (``PARRM._answer``: one C call, ``parrm_nm_minimise_fit``) sends back
    ``(results, log)``; one that cannot -- the batched driver, which merges the steps of several searches --
    sends back ``None`` and gets the same refinement as a sequence of :class:`FitRequest` steps."""

    __slots__ = ("starts", "y", "d_idx", "bandwidth", "lambda_")
    is_grid = False

    def __init__(self, starts, y, d_idx, bandwidth, lambda_):
        self.starts, self.y, self.d_idx, self.bandwidth, self.lambda_ = starts, y, d_idx, bandwidth, lambda_


def _fmin_requests(starts, y, d_idx, bandwidth, lambda_, log):
    """Lock-step Nelder-Mead from ``starts`` as a generator of :class:`FitRequest` (one per optimiser
    step); returns ``fmin_lockstep``'s result list.  Every evaluation is appended to ``log``."""
    whole = yield NelderMeadRequest([float(x) for x in starts], y, d_idx, bandwidth, lambda_)
    if whole is not None:
        results, evals = whole
        log.extend(evals)
        return results
    steps = fmin_lockstep_requests(starts)
    try:
        points = next(steps)
        while True:
            errors = yield FitRequest(y, d_idx, points, bandwidth, lambda_, False)
            log.append((points, errors))
            points = steps.send(errors)
    except StopIteration as stop:
        return stop.value


def _refine_candidates(periods: np.ndarray, errors: np.ndarray, y, d_idx, bandwidth, log):
    """Nelder-Mead (SciPy ``fmin`` defaults) from the best <= 5 candidates; keep the overall
    arg-min (parrm.py:467-522).  ``errors`` may be longer than ``periods`` (non-finite tail).
    The starts are independent in the reference too (an ordered map, :510-517); here they advance
    in lock-step so that each device batch serves all of them (``_neldermead``).  Generator: see
    :func:`_fmin_requests`; returns the 1-tuple holding the refined estimate."""
    n_starts = int(np.min((5, periods.shape[0])))
    results = yield from _fmin_requests([periods[i] for i in range(n_starts)], y, d_idx, bandwidth,
                                        _STAGE_LAMBDA, log)
    for i, (xopt, fopt, _, _) in enumerate(results):
        periods[i] = xopt[0]
        errors[i] = fopt
    return (periods[errors.argmin()],)


def find_period_batched(parrms, search_samples=None, assumed_periods=None, outlier_boundary=3.0,
                        random_seed=None, n_jobs=1) -> None:
    """``find_period`` for several ``PARRM`` objects at once -- per-site period estimation
    (examples/plot_example_dbs_data.py:52-98 builds one ``PARRM`` per recording site and calls
    ``find_period()`` on each): same arguments as :meth:`PARRM.find_period`, applied to every object.

    The searches advance together.  Each one's stage grids go to the device one after another (they
    fill the chip on their own), but the Nelder-Mead steps -- ~90 dependent, latency-bound evaluations
    per search -- of ALL searches are answered by one ``parrm_fit_errors_multi`` call per step, their
    small kernels overlapping on side streams.  Every evaluation is the single-instance evaluation
    (same kernels, same shapes), so each object ends with exactly the period its own
    ``find_period()`` would have found."""
    parrms = list(parrms)
    for p in parrms:
        if not isinstance(p, PARRM):
            raise TypeError("`parrms` must be PARRM objects.")
        p._say("\nFinding the artefact period...")
        p._reset_result_attrs()
        p._check_sort_find_stim_period_inputs(search_samples, assumed_periods, outlier_boundary, random_seed, n_jobs)
        p._in_call = False
        if not p.cache_on_device:
            p._d_data = p._d_data_src = None
    try:
        searches, pending, spaces = [], [], []
        grid_ws = _hip.FitWorkspace()
        for p in parrms:
            p._standardise_data()
            search = p._search_requests()
            searches.append(search)
            pending.append(next(search))
            spaces.append(_hip.FitWorkspace())

        def advance(search, answer):
            """Next FitRequest of a search (whole-refinement offers are declined: the steps of all searches are merged)."""
            request = search.send(answer)
            while isinstance(request, NelderMeadRequest):
                request = search.send(None)
            return request

        active = list(range(len(parrms)))
        while active:
            answers = {}
            # what goes into the shared call: optimiser steps, and stage grids small enough to be
            # latency-bound themselves (few channels, a few hundred candidates) -- unless the object
            # evaluates its grids its own way (ShardedPARRM's candidate slices)
            steps = [i for i in active
                     if not pending[i].is_grid
                     or (pending[i].periods.size <= _SMALL_GRID and type(parrms[i])._grid_errors is PARRM._grid_errors)]
            for i in active:
                if i not in steps:
                    answers[i] = parrms[i]._answer(pending[i], grid_ws)
            if steps:
                for i, errors in zip(steps, _hip.fit_errors_multi([(pending[i], spaces[i]) for i in steps])):
                    answers[i] = errors
            still = []
            for i in active:
                try:
                    pending[i] = advance(searches[i], answers[i])
                    still.append(i)
                except StopIteration:
                    pass
            active = still
    finally:
        for p in parrms:
            p._end_call()
    for p in parrms:
        p._say("    ... Artefact period found\n")
