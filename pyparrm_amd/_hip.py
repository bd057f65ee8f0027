"""ctypes binding of the gfx950 C-ABI library (``include/parrm_hip.h``) plus the torch plumbing
(device buffers, current stream) the façade needs.

There is deliberately no CPU fallback here: if the library is missing or no GPU is visible,
every compute entry point raises.  PyTorch is used only to own device memory and streams.
"""

from __future__ import annotations

import atexit
import ctypes as C
import os
import threading
import weakref

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libparrm_hip.so")

F32, F64 = 0, 1
KERNEL_AUTO, KERNEL_GATHER, KERNEL_STRIDE, KERNEL_PHASE, KERNEL_SEGMENTED = 0, 1, 2, 3, 4

# every symbol include/parrm_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "parrm_hip_abi_version",
    "parrm_hip_last_error",
    "parrm_hip_device_count",
    "parrm_hip_shutdown",
    "parrm_filter_plan_create",
    "parrm_filter_plan_destroy",
    "parrm_filter_plan_query",
    "parrm_filter_plan_set_kernel",
    "parrm_filter_apply",
    "parrm_filter_apply_window",
    "parrm_filter_apply_block",
    "parrm_filter_host",
    "parrm_filter_comb_precompile",
    "parrm_filter_plan_generated",
    "parrm_filter_plan_set_background",
    "parrm_filter_kernel_timing",
    "parrm_host_pin",
    "parrm_host_unpin",
    "parrm_absdiff_workspace_bytes",
    "parrm_absdiff_mean",
    "parrm_gather_standardise",
    "parrm_fit_workspace_bytes",
    "parrm_fit_errors",
    "parrm_fit_slice_workspace_bytes",
    "parrm_fit_errors_slice",
    "parrm_fit_errors_host",
    "parrm_fit_errors_multi",
    "parrm_nm_create",
    "parrm_nm_destroy",
    "parrm_nm_next",
    "parrm_nm_feed",
    "parrm_nm_result",
    "parrm_nm_fit_workspace_bytes",
    "parrm_nm_minimise_fit",
    "parrm_nm_chain_stats",
    "parrm_nmcore_create",
    "parrm_nmcore_destroy",
    "parrm_nmcore_next",
    "parrm_nmcore_feed",
    "parrm_nmcore_result",
)


class PlanInfo(C.Structure):
    _fields_ = [
        ("half_width", C.c_int64),
        ("n_taps", C.c_int64),
        ("n_runs", C.c_int64),
        ("stride", C.c_int64),
        ("n_delta", C.c_int64),
        ("ring_len", C.c_int64),
        ("rows_per_fill", C.c_int64),
        ("block_threads", C.c_int32),
        ("kernel", C.c_int32),
        ("phase_stride", C.c_int32),
        ("phase_delta", C.c_int32),
        ("phase_guard", C.c_int32),
        ("phase_groups", C.c_int32),
        ("phase_rows", C.c_int32),
        ("phase_row_slots", C.c_int32),
        ("phase_residues", C.c_int32),
        ("reserved", C.c_int32),
    ]


class FitProblem(C.Structure):
    """``parrm_fit_problem`` of include/parrm_hip.h."""

    _fields_ = [
        ("d_y", C.c_void_p),
        ("ldy", C.c_int64),
        ("d_idx", C.c_void_p),
        ("n_idx", C.c_int64),
        ("n_chans", C.c_int64),
        ("h_periods", C.c_void_p),
        ("n_periods", C.c_int64),
        ("bw", C.c_int32),
        ("reserved", C.c_int32),
        ("lambda_", C.c_double),
        ("h_err", C.c_void_p),
        ("d_workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class HipLibraryError(RuntimeError):
    """The HIP extension is missing, unloadable, or a call into it failed."""


_lib = None
_lock = threading.Lock()

# bench.py sets this to a list: every filter launch then appends a (start, stop) pair of HIP
# events recorded on the launch's own stream, so the kernel's duration can be read after a sync
FILTER_LAUNCH_EVENTS = None
# same for the candidate grids of find_period (calls of >= 64 candidates): entries are
# (start, stop, n_idx, n_chans, n_candidates, bandwidth)
FIT_GRID_EVENTS = None


def library_path() -> str:
    return _LIB_PATH


def lib() -> C.CDLL:
    """Load ``libparrm_hip.so`` (once) and declare its signatures."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise HipLibraryError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C pyparrm_amd/csrc`. pyparrm_amd has no CPU fallback."
            )
        # torch bundles its own HIP runtime (soname libamdhip64.so.7, the one this library
        # links against): import it first so both share ONE runtime; loading ours first would
        # pull in /opt/rocm's copy and leave the process with two.
        import torch  # noqa: F401

        try:
            L = C.CDLL(_LIB_PATH)
        except OSError as exc:  # e.g. libamdhip64 not found
            raise HipLibraryError(f"cannot load {_LIB_PATH}: {exc}") from exc
        i64, vp, dbl, i32 = C.c_int64, C.c_void_p, C.c_double, C.c_int
        L.parrm_hip_abi_version.restype = i32
        L.parrm_hip_abi_version.argtypes = []
        L.parrm_hip_last_error.restype = C.c_char_p
        L.parrm_hip_last_error.argtypes = []
        L.parrm_hip_device_count.restype = i32
        L.parrm_hip_device_count.argtypes = [C.POINTER(i32)]
        L.parrm_hip_shutdown.restype = i32
        L.parrm_hip_shutdown.argtypes = []
        L.parrm_filter_plan_create.restype = i32
        L.parrm_filter_plan_create.argtypes = [vp, i64, C.POINTER(vp)]
        L.parrm_filter_plan_destroy.restype = i32
        L.parrm_filter_plan_destroy.argtypes = [vp]
        L.parrm_filter_plan_query.restype = i32
        L.parrm_filter_plan_query.argtypes = [vp, C.POINTER(PlanInfo)]
        L.parrm_filter_plan_set_kernel.restype = i32
        L.parrm_filter_plan_set_kernel.argtypes = [vp, i32]
        L.parrm_filter_plan_generated.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
        L.parrm_filter_plan_set_background.restype = i32
        L.parrm_filter_plan_set_background.argtypes = [vp, i32]
        L.parrm_filter_kernel_timing.argtypes = [i32, C.POINTER(C.c_float)]
        L.parrm_filter_comb_precompile.argtypes = [vp, i64, i64, C.c_char_p, C.c_char_p, C.c_size_t]
        L.parrm_filter_apply.restype = i32
        L.parrm_filter_apply.argtypes = [vp, vp, i32, vp, i32, i64, i64, i64, i64, vp]
        L.parrm_filter_apply_window.restype = i32
        L.parrm_filter_apply_window.argtypes = [vp, vp, i32, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, vp]
        L.parrm_filter_apply_block.restype = i32
        L.parrm_filter_apply_block.argtypes = [vp, vp, i32, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp]
        L.parrm_filter_host.restype = i32
        L.parrm_filter_host.argtypes = [vp, vp, i32, vp, i32, i64, i64, i64, i64, i64]
        L.parrm_host_pin.restype = i32
        L.parrm_host_pin.argtypes = [vp, C.c_size_t]
        L.parrm_host_unpin.restype = i32
        L.parrm_host_unpin.argtypes = [vp]
        L.parrm_absdiff_workspace_bytes.restype = C.c_size_t
        L.parrm_absdiff_workspace_bytes.argtypes = [i64, i64]
        L.parrm_absdiff_mean.restype = i32
        L.parrm_absdiff_mean.argtypes = [vp, i32, i64, i64, i64, vp, vp, C.c_size_t, vp]
        L.parrm_gather_standardise.restype = i32
        L.parrm_gather_standardise.argtypes = [vp, i32, i64, i64, i64, vp, i64, vp, dbl, vp, i64, vp]
        L.parrm_fit_workspace_bytes.restype = C.c_size_t
        L.parrm_fit_workspace_bytes.argtypes = [i64, i64, i64, i32]
        L.parrm_fit_errors.restype = i32
        L.parrm_fit_errors.argtypes = [vp, i64, vp, i64, i64, vp, i64, i32, dbl, vp, vp, C.c_size_t, vp]
        L.parrm_fit_slice_workspace_bytes.restype = C.c_size_t
        L.parrm_fit_slice_workspace_bytes.argtypes = [i64, i64, i64, i64, i32]
        L.parrm_fit_errors_slice.restype = i32
        L.parrm_fit_errors_slice.argtypes = [vp, i64, vp, i64, i64, vp, i64, i64, i32, dbl, vp, vp, C.c_size_t, vp]
        L.parrm_fit_errors_host.restype = i32
        L.parrm_fit_errors_host.argtypes = [vp, i64, vp, i64, i64, vp, i64, i32, dbl, vp, vp, C.c_size_t, vp]
        L.parrm_fit_errors_multi.restype = i32
        L.parrm_fit_errors_multi.argtypes = [C.POINTER(FitProblem), i32, vp]
        L.parrm_nm_create.restype = i32
        L.parrm_nm_create.argtypes = [vp, i32, dbl, dbl, i32, i32, i32, C.POINTER(vp)]
        L.parrm_nm_destroy.restype = i32
        L.parrm_nm_destroy.argtypes = [vp]
        L.parrm_nm_next.restype = i32
        L.parrm_nm_next.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.parrm_nm_feed.restype = i32
        L.parrm_nm_feed.argtypes = [vp, vp, i32]
        L.parrm_nm_result.restype = i32
        L.parrm_nm_result.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(i32), C.POINTER(i32)]
        L.parrm_nm_chain_stats.restype = i32
        L.parrm_nm_chain_stats.argtypes = [vp]
        L.parrm_nmcore_create.restype = i32
        L.parrm_nmcore_create.argtypes = [vp, i32, dbl, dbl, i32, i32, i32, C.POINTER(vp)]
        L.parrm_nmcore_destroy.restype = i32
        L.parrm_nmcore_destroy.argtypes = [vp]
        L.parrm_nmcore_next.restype = i32
        L.parrm_nmcore_next.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.parrm_nmcore_feed.restype = i32
        L.parrm_nmcore_feed.argtypes = [vp, vp, i32]
        L.parrm_nmcore_result.restype = i32
        L.parrm_nmcore_result.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(i32), C.POINTER(i32)]
        L.parrm_nm_fit_workspace_bytes.restype = C.c_size_t
        L.parrm_nm_fit_workspace_bytes.argtypes = [i64, i64, i32]
        L.parrm_nm_minimise_fit.restype = i32
        L.parrm_nm_minimise_fit.argtypes = [vp, vp, i64, vp, i64, i64, i32, dbl, vp, C.c_size_t, vp, vp, vp, i32, vp, i32,
                                            C.POINTER(i32)]
        if L.parrm_hip_abi_version() != 3:
            raise HipLibraryError("libparrm_hip.so has an unexpected ABI version (rebuild it: make -C pyparrm_amd/csrc)")
        _lib = L
        # library-lifetime resources go back while the HIP runtime is certainly up: atexit hooks run at
        # the start of interpreter shutdown, before any module (torch, this one) is torn down
        atexit.register(_shutdown)
    return _lib


_SHUT_DOWN = False
dbl_t = C.c_double
_LIVE_PLANS: set = set()  # weak references to every FilterPlan alive


def _shutdown() -> None:
    global _SHUT_DOWN
    # every live plan frees its device tables NOW, while the runtime is up: a plan still held by a PARRM object
    # or an OnlineFilter would otherwise reach hipFree from __del__ during interpreter finalisation
    for ref in list(_LIVE_PLANS):
        plan = ref()
        if plan is not None:
            plan._release()
    _LIVE_PLANS.clear()
    _PLAN_CACHE.clear()
    _INDEX_STAGING.clear()
    if _lib is not None:
        _lib.parrm_hip_shutdown()
    _SHUT_DOWN = True


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().parrm_hip_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what or 'parrm_hip'} failed (code {rc}): {msg}")


_gpu_checked = None  # the torch module once a usable HIP device has been seen


def require_gpu():
    """Return the torch module after making sure a HIP device is usable; raise loudly otherwise."""
    global _gpu_checked
    if _gpu_checked is not None:
        return _gpu_checked
    L = lib()
    import torch

    n = C.c_int(0)
    check(L.parrm_hip_device_count(C.byref(n)), "parrm_hip_device_count")
    if n.value == 0 or not torch.cuda.is_available():
        raise HipLibraryError(
            "no MI355X/HIP device is visible: pyparrm_amd runs find_period/filter_data on the GPU "
            "only (there is no CPU fallback)."
        )
    _gpu_checked = torch
    return torch


def _stream_ptr(torch) -> int:
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # no Stream object per call
    if raw is not None:
        return int(raw(torch.cuda.current_device()))
    return int(torch.cuda.current_stream().cuda_stream)


class _on_device:
    """``with torch.cuda.device(dev)`` that costs nothing when ``dev`` is already current (the
    optimiser makes ~100 small calls per ``find_period``)."""

    __slots__ = ("_ctx",)

    def __init__(self, torch, device):
        index = device.index
        self._ctx = None if index is None or index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def _dtype_code(t) -> int:
    import torch

    if t.dtype == torch.float64:
        return F64
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported device dtype {t.dtype}; use float32 or float64")


def _check_rows(t, name: str) -> None:
    if t.dim() != 2 or not t.is_cuda:
        raise ValueError(f"`{name}` must be a 2-D CUDA tensor")
    if t.shape[1] > 0 and t.stride(1) != 1:
        raise ValueError(f"`{name}` must be contiguous along time")


class FilterPlan:
    """Device-side plan for one PARRM filter (``parrm_filter_plan_*``)."""

    def __init__(self, filt: np.ndarray, device: int | None = None):
        """``device``: index of the HIP device whose memory holds the plan's tables (default: the
        current device).  A plan only serves recordings on that device."""
        self._h = C.c_void_p(None)
        torch = require_gpu()
        f = np.ascontiguousarray(filt, dtype=np.float64)
        if f.ndim != 1:
            raise ValueError("filter must be 1-D")
        self.device = torch.cuda.current_device() if device is None else int(device)
        with torch.cuda.device(self.device):
            check(
                lib().parrm_filter_plan_create(f.ctypes.data_as(C.c_void_p), f.shape[0], C.byref(self._h)),
                "parrm_filter_plan_create",
            )
        _LIVE_PLANS.add(weakref.ref(self, _LIVE_PLANS.discard))

    def _refuse_copy(self, *_args):
        raise TypeError("a FilterPlan owns device memory and cannot be copied or pickled; build one "
                        "from the filter array")

    __copy__ = __deepcopy__ = __reduce__ = _refuse_copy

    def _check_device(self, t) -> None:
        if t.device.index != self.device:
            raise ValueError(
                f"this FilterPlan lives on cuda:{self.device} but the recording is on {t.device}; "
                "build a plan for that device"
            )

    def _release(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value and _lib is not None and not _SHUT_DOWN:
            _lib.parrm_filter_plan_destroy(h)

    def __del__(self):
        self._release()

    @property
    def info(self) -> PlanInfo:
        out = PlanInfo()
        check(lib().parrm_filter_plan_query(self._h, C.byref(out)), "parrm_filter_plan_query")
        return out

    @property
    def generated(self):
        """(state, stride, message) of the per-filter generated kernel: state 1 = in use, 0 = not tried yet, -1 = unavailable."""
        state, stride = C.c_int(0), C.c_int(0)
        msg = C.create_string_buffer(512)
        check(lib().parrm_filter_plan_generated(self._h, C.byref(state), C.byref(stride), msg, 512), "filter_plan_generated")
        return state.value, stride.value, msg.value.decode(errors="replace")

    def set_kernel(self, kernel: int) -> None:
        check(lib().parrm_filter_plan_set_kernel(self._h, kernel), "parrm_filter_plan_set_kernel")

    def set_background_compile(self, on: bool = True) -> None:
        """A generated kernel that has to be compiled first (~1.7 s of hipRTC for a geometry nobody has built yet) is
        built by a worker thread while the generic kernels serve the launches; ``generated`` reports state 2 meanwhile."""
        check(lib().parrm_filter_plan_set_background(self._h, 1 if on else 0), "parrm_filter_plan_set_background")

    def wait_generated(self, timeout: float = 60.0):
        """Block until a background build has ended (or ``timeout`` seconds); returns ``generated``."""
        import time

        t0 = time.monotonic()
        while True:
            g = self.generated
            if g[0] != 2 or time.monotonic() - t0 > timeout:
                return g
            time.sleep(0.01)

    def apply(self, x, out=None, out_dtype=None, total_chans=None):
        """y = filter(x) for a device-resident recording ``x[C, N]`` (f32/f64).  ``total_chans``: ``x`` is
        a channel block of a recording with that many channels; the time axis is then cut as for the whole
        recording, so the block's output is bit-identical to the same rows of an unsharded call."""
        torch = require_gpu()
        _check_rows(x, "x")
        self._check_device(x)
        if out is None:
            out = torch.empty(x.shape, dtype=out_dtype or torch.float64, device=x.device)
        _check_rows(out, "out")
        self._check_device(out)
        if out.shape != x.shape:
            raise ValueError("`out` must have the shape of `x`")
        n_chans, n_samples = x.shape
        if n_chans == 0 or n_samples == 0:
            return out
        ldx = x.stride(0) if n_chans > 1 else max(x.stride(0), n_samples)
        ldy = out.stride(0) if n_chans > 1 else max(out.stride(0), n_samples)
        with torch.cuda.device(x.device):
            events = None
            if FILTER_LAUNCH_EVENTS is not None:
                events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events[0].record()  # torch's current stream == the stream passed to the launch
            check(
                lib().parrm_filter_apply_block(
                    self._h, x.data_ptr(), _dtype_code(x), out.data_ptr(), _dtype_code(out),
                    n_chans, max(int(total_chans or 0), n_chans), 0, n_samples, 0, n_samples, n_samples,
                    ldx, ldy, _stream_ptr(torch),
                ),
                "parrm_filter_apply_block",
            )
            if events is not None:
                events[1].record()
                FILTER_LAUNCH_EVENTS.append(events)
        return out

    def apply_window(self, x, buf_first: int, out_first: int, out_len: int, n_total: int, out=None):
        """Outputs [out_first, out_first+out_len) of a recording of ``n_total`` samples, of which
        ``x[C, buf_len]`` holds samples [buf_first, buf_first+buf_len)."""
        torch = require_gpu()
        _check_rows(x, "x")
        self._check_device(x)
        if out is None:
            out = torch.empty((x.shape[0], out_len), dtype=torch.float64, device=x.device)
        self._check_device(out)
        with torch.cuda.device(x.device):
            check(
                lib().parrm_filter_apply_window(
                    self._h, x.data_ptr(), _dtype_code(x), out.data_ptr(), _dtype_code(out), x.shape[0],
                    buf_first, x.shape[1], out_first, out_len, n_total,
                    max(x.stride(0), x.shape[1]), max(out.stride(0), out_len), _stream_ptr(torch),
                ),
                "parrm_filter_apply_window",
            )
        return out

    def apply_host(self, x: np.ndarray, out_dtype=np.float64, chunk_samples: int = 0, out=None) -> np.ndarray:
        """Stream a host-resident recording through the device in time chunks.

        ``out`` (optional) receives the result; pass arrays locked with :func:`pin_host` when the
        same buffers are filtered repeatedly -- page-locking costs more than the transfer."""
        require_gpu()
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        x = np.ascontiguousarray(x)
        if out is None:
            y = np.empty(x.shape, dtype=out_dtype)
        else:
            y = out
            if y.shape != x.shape or not y.flags.c_contiguous or y.dtype not in (np.float32, np.float64):
                raise ValueError("`out` must be a C-contiguous float32/float64 array of the input's shape")
        code_x = F64 if x.dtype == np.float64 else F32
        code_y = F64 if y.dtype == np.float64 else F32
        torch = require_gpu()
        with torch.cuda.device(self.device):
            self._filter_host(x, code_x, y, code_y, chunk_samples)
        return y

    def _filter_host(self, x, code_x, y, code_y, chunk_samples):
        check(
            lib().parrm_filter_host(
                self._h, x.ctypes.data_as(C.c_void_p), code_x, y.ctypes.data_as(C.c_void_p), code_y,
                x.shape[0], x.shape[1], x.shape[1], x.shape[1], chunk_samples,
            ),
            "parrm_filter_host",
        )
        return y


_PLAN_CACHE: dict = {}
_PLAN_CACHE_SIZE = 16
_PLAN_CACHE_LOCK = threading.Lock()


def filter_kernel_timing(enable: bool) -> float:
    """Switch the library's own event bracket around the MAIN filter kernel on or off for this thread; returns the
    duration in ms of the most recent bracketed kernel (-1.0 when there is none)."""
    ms = C.c_float(-1.0)
    check(lib().parrm_filter_kernel_timing(1 if enable else 0, C.byref(ms)), "filter_kernel_timing")
    return float(ms.value)


def precompile_filter_kernel(filt: np.ndarray, out_dir: str | None = None, stride: int = 0) -> str:
    """Generate and compile (hipRTC; no GPU needed) the per-filter kernel of `filt` -- a filter array as
    `PARRM.create_filter` makes them -- into `out_dir` (default: the in-tree `lib/kernels/`, which the library
    searches before its user cache) and return the path of the generated source.  Raises `HipLibraryError`
    when the generated form does not take the filter."""
    out_dir = out_dir or os.path.join(os.path.dirname(library_path()), "kernels")
    filt = np.ascontiguousarray(filt, dtype=np.float64)
    buf = C.create_string_buffer(4096)
    check(lib().parrm_filter_comb_precompile(filt.ctypes.data, filt.size, stride, out_dir.encode(), buf, 4096),
          "filter_comb_precompile")
    return buf.value.decode()


def shared_filter_plan(filt: np.ndarray, device: int) -> FilterPlan:
    """Plan for ``filt`` on ``device`` from a small process-wide cache keyed by the filter's content: a
    plan is immutable (tables on the device), building one costs a stride search, a ``hipMalloc`` and a
    synchronous upload, and the same filter comes back often -- the reference's explorer revisits
    settings per widget event (_utils/_plotting.py:568-584), several ``PARRM`` objects of one study share
    a period.  Plans with a forced kernel variant (tests) are never shared: this returns AUTO plans only."""
    f = np.ascontiguousarray(filt, dtype=np.float64)
    key = (int(device), f.shape[0], hash(f.tobytes()))
    with _PLAN_CACHE_LOCK:
        hit = _PLAN_CACHE.get(key)
        if hit is not None and np.array_equal(hit[0], f):
            _PLAN_CACHE[key] = _PLAN_CACHE.pop(key)  # most recently used last
            return hit[1]
    plan = FilterPlan(f, device=device)  # (outside the lock: a stride search, a hipMalloc and an upload)
    with _PLAN_CACHE_LOCK:
        hit = _PLAN_CACHE.get(key)
        if hit is not None and np.array_equal(hit[0], f):  # another thread built it meanwhile
            return hit[1]
        _PLAN_CACHE[key] = (f.copy(), plan)
        while len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    return plan


PINNED_OUTPUT_BYTES = 16 << 30  # results up to this size come back through page-locked memory


def to_host_numpy(t) -> np.ndarray:
    """Device tensor -> NumPy array.  The copy lands in page-locked memory owned by PyTorch's pinned
    allocator (the array keeps it alive): 50-57 GB/s instead of the 6-13 GB/s of a pageable
    ``.cpu()`` -- for NumPy-in/NumPy-out callers the read-back, not the kernel, is the cost of
    ``filter_data``."""
    torch = require_gpu()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0 or nbytes > PINNED_OUTPUT_BYTES:
        return t.cpu().numpy()
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:  # the page-lock limit of the host: fall back to the pageable copy
        return t.cpu().numpy()
    # a BLOCKING copy: a non-blocking one makes PyTorch's pinned allocator tie the block to the stream
    # and record an event when the array is finally dropped -- at interpreter exit that lands in a
    # runtime that may already be gone (seen as an intermittent crash after a green test run)
    host.copy_(t)
    return host.numpy()


def pin_host(array: np.ndarray) -> None:
    """Page-lock a host array for :meth:`FilterPlan.apply_host` (no-op when it already is)."""
    require_gpu()
    check(lib().parrm_host_pin(array.ctypes.data_as(C.c_void_p), array.nbytes), "parrm_host_pin")


def unpin_host(array: np.ndarray) -> None:
    """Undo :func:`pin_host` (no-op for memory that is not page-locked)."""
    require_gpu()
    check(lib().parrm_host_unpin(array.ctypes.data_as(C.c_void_p)), "parrm_host_unpin")


_NO_LEAN_CALLS = bool(os.environ.get("PARRM_NO_LEAN_CALLS"))  # A/B knob: the generic wrapper for every call
_INDEX_STAGING: dict = {}
_STAGING_LOCK = threading.Lock()  # the staging vector is filled and copied from under this lock (two
#                                   threads searching on one device overwrote each other's indices mid-copy)


def upload_indices(indices: np.ndarray, device):
    """Stage sample indices (ascending int64) as a device vector.  A contiguous range -- stages 1 and 2 of the
    period search (parrm.py:352-365) -- is generated on the device; anything else goes through a page-locked
    staging vector kept per device (a pageable copy of 200 KB costs ~1 ms, three times per search)."""
    torch = require_gpu()
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    n = idx.shape[0]
    if n and int(idx[-1]) - int(idx[0]) + 1 == n and (n < 3 or bool(np.all(idx[1:] - idx[:-1] == 1))):
        return torch.arange(int(idx[0]), int(idx[0]) + n, dtype=torch.int64, device=device)
    key = torch.device(device).index
    with _STAGING_LOCK:
        staging = _INDEX_STAGING.get(key)
        if staging is None or staging.shape[0] < n:
            staging = torch.empty(max(n, 32768), dtype=torch.int64).pin_memory()
            _INDEX_STAGING[key] = staging
        staging[:n] = torch.from_numpy(idx)
        return staging[:n].to(device)  # blocking: the staging vector is free again when the lock is released


def absdiff_mean(x):
    """scale[c] = mean_i |x[c,i+1]-x[c,i]| on the device (parrm.py:274-275)."""
    torch = require_gpu()
    _check_rows(x, "x")
    n_chans, n_samples = x.shape
    ws_bytes = lib().parrm_absdiff_workspace_bytes(n_chans, n_samples)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x.device)
    scale = torch.empty(n_chans, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(
            lib().parrm_absdiff_mean(
                x.data_ptr(), _dtype_code(x), n_chans, n_samples, max(x.stride(0), n_samples),
                scale.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr(torch),
            ),
            "parrm_absdiff_mean",
        )
    return scale


def gather_standardise(x, idx, scale, outlier_boundary: float):
    """Y[j, c] = clip((x[c,idx_j+1]-x[c,idx_j])/scale[c], +-ob); returns a [n_idx, C] f64 tensor."""
    torch = require_gpu()
    _check_rows(x, "x")
    n_chans, n_samples = x.shape
    # rows padded to whole column quads (zeros): the matrix-core Gram kernel reads 16-byte quads
    ld = (n_chans + 3) // 4 * 4
    y = (torch.zeros if ld != n_chans else torch.empty)((idx.shape[0], ld), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(
            lib().parrm_gather_standardise(
                x.data_ptr(), _dtype_code(x), n_chans, n_samples, max(x.stride(0), n_samples),
                idx.data_ptr(), idx.shape[0], scale.data_ptr(), float(outlier_boundary),
                y.data_ptr(), ld, _stream_ptr(torch),
            ),
            "parrm_gather_standardise",
        )
    return y[:, :n_chans]


class FitWorkspace:
    """Grow-only device scratch for ``parrm_fit_errors`` plus pinned staging buffers for the
    candidate periods / errors (one per stage; avoids per-call mallocs and pageable copies --
    the Nelder-Mead phase is ~100 small launches whose cost is mostly this plumbing)."""

    MAX_BYTES = 12 << 30

    #: candidates per call the lean optimiser-step path takes (``small_batch``)
    LEAN_MAX = 64

    def __init__(self):
        self._buf = None
        self._sizes = {}
        self._h_per = self._h_err = self._d_per = self._d_err = None
        self._lean = None  # (y, idx, bandwidth) and everything derived from them, see small_batch

    def small_batch(self, torch, y, idx, periods: np.ndarray, bandwidth: int, lambda_: float):
        """One optimiser step (<= ``LEAN_MAX`` abscissae) through ``parrm_fit_errors_host`` with everything that
        does not change between the steps of a stage -- pointers, strides, sizes, the staging arrays' addresses --
        prepared once: the generic wrapper costs ~25 us of Python per call, a Nelder-Mead phase makes ~80 of them."""
        lean = self._lean
        if lean is None or lean[0] is not y or lean[1] is not idx or lean[2] != bandwidth:
            n_idx, n_chans = y.shape
            h_per, h_out = np.empty(self.LEAN_MAX, dtype=np.float64), np.empty(self.LEAN_MAX, dtype=np.float64)
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            lean = [y, idx, bandwidth, y.data_ptr(), y.stride(0), idx.data_ptr(), int(n_idx), int(n_chans), y.device,
                    y.device.index, h_per, h_out, h_per.ctypes.data, h_out.ctypes.data, {}, raw,
                    lib().parrm_fit_errors_host]
            self._lean = lean
        (_, _, _, y_ptr, ldy, idx_ptr, n_idx, n_chans, device, dev_index, h_per, h_out, per_addr, out_addr, sizes, raw,
         call) = lean
        n_per = periods.shape[0]
        h_per[:n_per] = periods
        nbytes = sizes.get(n_per)
        if nbytes is None:
            nbytes = sizes[n_per] = self.nbytes(n_idx, n_chans, n_per, bandwidth) + 16 * n_per
        buf = self._buf
        if buf is None or buf.numel() * 8 < nbytes or buf.device != device:
            buf = self.get(nbytes, device)
        if dev_index != torch.cuda.current_device():
            with torch.cuda.device(device):
                rc = call(y_ptr, ldy, idx_ptr, n_idx, n_chans, per_addr, n_per, bandwidth, lambda_, out_addr,
                          buf.data_ptr(), nbytes, _stream_ptr(torch))
        else:
            stream = int(raw(dev_index)) if raw is not None else _stream_ptr(torch)
            rc = call(y_ptr, ldy, idx_ptr, n_idx, n_chans, per_addr, n_per, bandwidth, lambda_, out_addr,
                      buf.data_ptr(), nbytes, stream)
        if rc:
            check(rc, "parrm_fit_errors_host")
        return h_out[:n_per].copy()

    def get(self, nbytes: int, device):
        torch = require_gpu()
        if self._buf is None or self._buf.numel() * 8 < nbytes or self._buf.device != device:
            self._buf = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)
        return self._buf

    def nbytes(self, n_idx: int, n_chans: int, n_per: int, bandwidth: int, grid: int = 0) -> int:
        key = (n_idx, n_chans, n_per, bandwidth, grid)
        if key not in self._sizes:
            self._sizes[key] = lib().parrm_fit_slice_workspace_bytes(n_idx, n_chans, n_per, max(grid, n_per), bandwidth)
        return self._sizes[key]

    def staging(self, n_per: int, device):
        torch = require_gpu()
        if self._h_per is None or self._h_per.numel() < n_per or self._d_per.device != device:
            cap = max(64, 2 * n_per)
            self._h_per = torch.empty(cap, dtype=torch.float64).pin_memory()
            self._h_err = torch.empty(cap, dtype=torch.float64).pin_memory()
            self._d_per = torch.empty(cap, dtype=torch.float64, device=device)
            self._d_err = torch.empty(cap, dtype=torch.float64, device=device)
        return self._h_per, self._h_err, self._d_per, self._d_err


_NM_WORKSPACE_BYTES: dict = {}


class NativeNelderMead:
    """The lock-step Nelder-Mead of ``pyparrm_amd/_neldermead.py`` in the C library (``csrc/parrm_nm.hip``): the same
    batches, the same decisions.  Step interface (``next_batch`` / ``feed``: needs no GPU -- tests drive it against
    the Python generator) and the whole search on the device objective in one C call (``minimise_fit``)."""

    MAX_BATCH = 64
    _FAMILY = "parrm_nm"  # the entry points' prefix (NelderMeadCoreOnHost: the state machine the device runs)

    def _fn(self, name):
        return getattr(lib(), f"{self._FAMILY}_{name}")

    def __init__(self, starts, xtol=1e-4, ftol=1e-4, maxiter=None, maxfun=None, lookahead_runs=2):
        x0 = np.ascontiguousarray(np.asarray(starts, dtype=np.float64).reshape(-1))
        self._n = int(x0.shape[0])
        h = C.c_void_p()
        check(self._fn("create")(x0.ctypes.data, self._n, float(xtol), float(ftol), -1 if maxiter is None else int(maxiter),
                                 -1 if maxfun is None else int(maxfun), int(lookahead_runs), C.byref(h)), f"{self._FAMILY}_create")
        self._h = h
        self._buf = np.empty(max(self.MAX_BATCH, 9 * self._n), dtype=np.float64)  # (3 per run + 6 of look-ahead)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and not _SHUT_DOWN and _lib is not None:  # (interpreter shutdown may have cleared the module already)
            try:
                getattr(_lib, f"{self._FAMILY}_destroy")(h)
            except Exception:
                pass

    def next_batch(self):
        """The next batch of abscissae (ascending float64 array), or None when every run has ended."""
        n = C.c_int32(0)
        check(self._fn("next")(self._h, self._buf.ctypes.data, int(self._buf.shape[0]), C.byref(n)), f"{self._FAMILY}_next")
        return self._buf[: n.value].copy() if n.value else None

    def feed(self, values) -> None:
        v = np.ascontiguousarray(values, dtype=np.float64)
        check(self._fn("feed")(self._h, v.ctypes.data, int(v.shape[0])), f"{self._FAMILY}_feed")

    def results(self):
        """``[(xopt[1], fopt, iterations, funcalls)]`` per start, as ``fmin_lockstep`` returns them."""
        out = []
        x, f, it, nf = dbl_t(), dbl_t(), C.c_int32(), C.c_int32()
        for r in range(self._n):
            check(self._fn("result")(self._h, r, C.byref(x), C.byref(f), C.byref(it), C.byref(nf)), f"{self._FAMILY}_result")
            out.append((np.array([x.value], dtype=np.float64), float(f.value), int(it.value), int(nf.value)))
        return out

    def minimise_fit(self, y, idx, bandwidth: int, lambda_: float, workspace: "FitWorkspace"):
        """Run the search to its end on the harmonic-regression objective of the stage matrix ``y`` / indices ``idx``
        (device tensors); returns ``(results, log)`` with ``log`` the list of ``(points, errors)`` per batch."""
        torch = require_gpu()
        if 9 * self._n > self.MAX_BATCH:
            # more starts than one C call's batch buffer takes (the reference starts <= 5, parrm.py:499): the same
            # refinement through the step interface, one device batch per step
            log = []
            while True:
                points = self.next_batch()
                if points is None:
                    return self.results(), log
                values = fit_errors(y, idx, points, bandwidth, lambda_, workspace)
                self.feed(values)
                log.append((points, np.asarray(values, dtype=np.float64)))
        n_idx, n_chans = (int(v) for v in y.shape)
        key = (n_idx, n_chans, int(bandwidth))
        nbytes = _NM_WORKSPACE_BYTES.get(key)
        if nbytes is None:  # (the library scans every batch size up to 64 for the largest scratch: once per shape)
            nbytes = _NM_WORKSPACE_BYTES[key] = int(lib().parrm_nm_fit_workspace_bytes(n_idx, n_chans, int(bandwidth)))
        if nbytes == 0:
            raise ValueError("bad shape for the fit objective")
        buf = workspace.get(nbytes, y.device)
        cap, bcap = 4096, 512
        hx, hf = np.empty(cap, dtype=np.float64), np.empty(cap, dtype=np.float64)
        sizes = np.empty(bcap, dtype=np.int32)
        nb = C.c_int32(0)
        with torch.cuda.device(y.device):
            check(lib().parrm_nm_minimise_fit(self._h, y.data_ptr(), y.stride(0), idx.data_ptr(), n_idx, n_chans, int(bandwidth),
                                              float(lambda_), buf.data_ptr(), nbytes, _stream_ptr(torch), hx.ctypes.data,
                                              hf.ctypes.data, cap, sizes.ctypes.data, bcap, C.byref(nb)),
                  "parrm_nm_minimise_fit")
        log, at = [], 0
        for k in range(nb.value):
            n = int(sizes[k])
            log.append((hx[at:at + n].copy(), hf[at:at + n].copy()))
            at += n
        return self.results(), log


class NelderMeadCoreOnHost(NativeNelderMead):
    """``csrc/parrm_nm_core.h`` -- the plain-data state machine the DEVICE runs between two optimiser batches
    (``nm_chain_step_kernel``) -- compiled for the host: step interface only (a test surface; needs no GPU)."""

    _FAMILY = "parrm_nmcore"
    MAX_BATCH = 32

    def minimise_fit(self, *args, **kwargs):
        raise NotImplementedError("the host build of the device's state machine has the step interface only")


def fit_errors(y, idx, periods: np.ndarray, bandwidth: int, lambda_: float, workspace: FitWorkspace | None = None,
               grid_periods: int = 0):
    """Channel-averaged regularised fit error of every candidate period (parrm.py:552-632).

    ``y`` is the [n_idx, C] matrix from :func:`gather_standardise`, ``idx`` the int64 device vector
    of sample indices.  Returns a float64 NumPy vector (one blocking read-back per call).

    ``grid_periods``: ``periods`` is a slice of a grid of that many candidates evaluated over several
    calls / GPUs; each candidate's error is then bit-identical to a single call on the whole grid
    (``parrm_fit_errors_slice``).
    """
    torch = require_gpu()
    periods = np.ascontiguousarray(np.asarray(periods, dtype=np.float64).reshape(-1))
    n_per = periods.shape[0]
    if n_per == 0:
        return np.empty(0, dtype=np.float64)
    if workspace is not None and n_per <= FitWorkspace.LEAN_MAX and not grid_periods and not _NO_LEAN_CALLS:
        return workspace.small_batch(torch, y, idx, periods, int(bandwidth), float(lambda_))
    L = lib()
    n_idx, n_chans = y.shape
    ws = workspace or FitWorkspace()
    if ws.nbytes(n_idx, n_chans, 1, bandwidth) == 0:
        raise HipLibraryError("parrm_fit_workspace_bytes rejected the problem shape")
    # batch so that the scratch stays bounded
    grid = int(grid_periods) if grid_periods and grid_periods > n_per else 0
    batch = min(n_per, 65535)
    while batch > 1 and ws.nbytes(n_idx, n_chans, batch, bandwidth, grid) > FitWorkspace.MAX_BYTES:
        batch = (batch + 1) // 2
    if batch != n_per and not grid:
        grid = n_per  # a grid cut only to bound the scratch is still planned as one grid
    if batch == n_per and not grid:
        # one fused call: periods in, kernels, errors out, stream sync (the optimiser's small batches)
        nbytes = ws.nbytes(n_idx, n_chans, n_per, bandwidth) + 16 * n_per
        buf = ws.get(nbytes, y.device)
        out = np.empty(n_per, dtype=np.float64)
        with _on_device(torch, y.device):
            events = None
            if FIT_GRID_EVENTS is not None and n_per >= 64:
                events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events[0].record()
            check(
                L.parrm_fit_errors_host(
                    y.data_ptr(), y.stride(0), idx.data_ptr(), n_idx, n_chans,
                    periods.ctypes.data_as(C.c_void_p), n_per, int(bandwidth), float(lambda_),
                    out.ctypes.data_as(C.c_void_p), buf.data_ptr(), nbytes, _stream_ptr(torch),
                ),
                "parrm_fit_errors_host",
            )
            if events is not None:
                events[1].record()
                FIT_GRID_EVENTS.append((events[0], events[1], int(n_idx), int(n_chans), int(n_per), int(bandwidth)))
        return out
    h_per, h_err, d_per, d_err = ws.staging(n_per, y.device)
    h_per[:n_per] = torch.from_numpy(periods)
    with torch.cuda.device(y.device):
        stream = _stream_ptr(torch)
        d_per[:n_per].copy_(h_per[:n_per], non_blocking=True)
        events = None
        if FIT_GRID_EVENTS is not None and n_per >= 64:
            events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            events[0].record()
        for lo in range(0, n_per, batch):
            cnt = min(batch, n_per - lo)
            nbytes = ws.nbytes(n_idx, n_chans, cnt, bandwidth, grid)
            buf = ws.get(nbytes, y.device)
            check(
                L.parrm_fit_errors_slice(
                    y.data_ptr(), y.stride(0), idx.data_ptr(), n_idx, n_chans,
                    d_per.data_ptr() + 8 * lo, cnt, max(grid, cnt), int(bandwidth), float(lambda_),
                    d_err.data_ptr() + 8 * lo, buf.data_ptr(), nbytes, stream,
                ),
                "parrm_fit_errors_slice",
            )
        if events is not None:
            events[1].record()
            FIT_GRID_EVENTS.append((events[0], events[1], int(n_idx), int(n_chans), int(n_per), int(bandwidth)))
        h_err[:n_per].copy_(d_err[:n_per], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return h_err[:n_per].numpy().copy()


MULTI_MAX_PROBLEMS, MULTI_MAX_PERIODS = 64, 4096


def fit_errors_multi(items):
    """Answer several optimiser steps of independent searches in one call (``parrm_fit_errors_multi``).

    ``items``: sequence of ``(request, workspace)`` -- ``request`` has ``y``, ``d_idx``, ``periods``,
    ``bandwidth``, ``lambda_`` (a ``parrm.FitRequest``), ``workspace`` is that search's own
    :class:`FitWorkspace` (the problems run concurrently, so they cannot share scratch).  Returns the
    list of float64 error vectors, each bit-identical to ``fit_errors`` on the same request."""
    torch = require_gpu()
    L = lib()
    items = list(items)
    out = [None] * len(items)
    lo = 0
    while lo < len(items):
        hi, total = lo, 0
        while hi < len(items) and hi - lo < MULTI_MAX_PROBLEMS:
            n = int(np.asarray(items[hi][0].periods).size)
            if n > MULTI_MAX_PERIODS:
                break
            if total + n > MULTI_MAX_PERIODS:
                break
            total += n
            hi += 1
        if hi == lo:  # a request too large for the hand-off block: the plain path
            req, ws = items[lo]
            out[lo] = fit_errors(req.y, req.d_idx, req.periods, req.bandwidth, req.lambda_, ws)
            lo += 1
            continue
        problems = (FitProblem * (hi - lo))()
        keep = []
        device = items[lo][0].y.device
        for k, (req, ws) in enumerate(items[lo:hi]):
            if req.y.device != device:
                raise ValueError("fit_errors_multi: all stage matrices of one call must be on one device")
            periods = np.ascontiguousarray(np.asarray(req.periods, dtype=np.float64).reshape(-1))
            n_idx, n_chans = req.y.shape
            n_per = periods.shape[0]
            nbytes = ws.nbytes(n_idx, n_chans, n_per, req.bandwidth) + 16 * n_per
            if nbytes == 16 * n_per:
                raise HipLibraryError("parrm_fit_workspace_bytes rejected the problem shape")
            buf = ws.get(nbytes, device)
            err = np.empty(n_per, dtype=np.float64)
            keep.append((periods, buf, err))
            q = problems[k]
            q.d_y, q.ldy, q.d_idx = req.y.data_ptr(), req.y.stride(0), req.d_idx.data_ptr()
            q.n_idx, q.n_chans = n_idx, n_chans
            q.h_periods, q.n_periods = periods.ctypes.data, n_per
            q.bw, q.reserved, q.lambda_ = int(req.bandwidth), 0, float(req.lambda_)
            q.h_err, q.d_workspace, q.workspace_bytes = err.ctypes.data, buf.data_ptr(), nbytes
            out[lo + k] = err
        with _on_device(torch, device):
            check(L.parrm_fit_errors_multi(problems, hi - lo, _stream_ptr(torch)), "parrm_fit_errors_multi")
        del keep
        lo = hi
    return out
