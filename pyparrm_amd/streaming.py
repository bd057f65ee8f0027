"""The callers either side of ``filter_data`` for recordings that do not sit in one array
(SURVEY.md section 8f-4):

* :class:`OnlineFilter` -- stateful block-by-block filtering: push consecutive blocks of samples as
  they arrive, get back every output that has become computable.  With a one-sided filter whose
  taps only reach EARLIER samples (the reference's ``filter_direction="future"``: it keeps the taps
  ``w > 0`` of ``y[n] = x[n] - mean x[n - w]``, parrm.py:819-820) the latency is zero samples; taps
  that reach later samples (``"past"`` keeps ``w <= 0``, :817-818; ``"both"``) delay the output by
  that reach.  The concatenation of everything returned equals ``filter_data`` of the concatenated
  signal (the edge divisors of parrm.py:862-866 included).
* :func:`filter_file` -- ``.npy`` in, ``.npy`` out through memory maps (the reference ships its
  recordings as ``.npy`` files, data/example_data.py:13-30): time chunks go from the mapped file into
  page-locked buffers, to the device, through ``parrm_filter_apply_window`` and back, two chunks in
  flight on two streams.  Nothing larger than two chunks is ever resident on the host or the device.

Both are host orchestration over the C ABI's window form; the arithmetic is the resident kernels'.
"""

from __future__ import annotations

import numpy as np

from . import _hip


def _tap_reach(filt: np.ndarray):
    """(half_width, reach into earlier samples, reach into later samples) of a dense PARRM filter
    (centre 1, taps -1/S at offsets w; tap w reads sample n - w)."""
    filt = np.asarray(filt, dtype=np.float64)
    if filt.ndim != 1 or filt.shape[0] < 3 or filt.shape[0] % 2 == 0:
        raise ValueError("`filt` must be a dense PARRM filter of odd length >= 3.")
    hw = (filt.shape[0] - 1) // 2
    w = np.flatnonzero(filt) - hw
    w = w[w != 0]
    if w.size == 0:
        raise ValueError("`filt` has no taps.")
    return hw, int(max(w.max(), 0)), int(max(-w.min(), 0))


class OnlineFilter:
    """Block-by-block ``filter_data``.

    Parameters
    ----------
    filt : numpy.ndarray
        Dense filter as ``PARRM.filter`` holds it (parrm.py:803-833).
    n_chans : int
        Channels per block.
    dtype : numpy dtype (default float64)
        Input sample type (float32 or float64); outputs are float64 (the reference's rule) unless
        ``out_dtype`` says float32 for float32 input.
    device : int | None
        HIP device index (default: the current one).
    """

    def __init__(self, filt, n_chans: int, dtype=np.float64, out_dtype=np.float64, device=None):
        torch = _hip.require_gpu()
        self._hw, self._back, self._ahead = _tap_reach(filt)
        if not isinstance(n_chans, int) or n_chans < 1:
            raise ValueError("`n_chans` must be a positive int.")
        dtype, out_dtype = np.dtype(dtype), np.dtype(out_dtype)
        if dtype not in (np.float32, np.float64) or out_dtype not in (np.float32, np.float64):
            raise TypeError("`dtype` and `out_dtype` must be float32 or float64.")
        if out_dtype == np.float32 and dtype != np.float32:
            raise TypeError("float32 output needs float32 input.")
        self._n_chans = n_chans
        self._t_in = torch.float64 if dtype == np.float64 else torch.float32
        self._t_out = torch.float64 if out_dtype == np.float64 else torch.float32
        self._index = torch.cuda.current_device() if device is None else int(device)
        self._plan = _hip.FilterPlan(filt, device=self._index)
        self._device = torch.device("cuda", self._index)
        self._buf = None        # device samples [first, first + held)
        self._first = 0         # global index of the buffer's first sample
        self._held = 0
        self._received = 0      # samples pushed so far
        self._emitted = 0       # outputs returned so far
        self._closed = False

    @property
    def latency(self) -> int:
        """Samples an output lags behind its input: the filter's reach into later samples."""
        return self._ahead

    @property
    def n_received(self) -> int:
        return self._received

    @property
    def n_emitted(self) -> int:
        return self._emitted

    def _append(self, block):
        torch = _hip.require_gpu()
        if isinstance(block, np.ndarray):
            if block.ndim != 2 or block.shape[0] != self._n_chans:
                raise ValueError("`block` must have shape [n_chans, times].")
            t = torch.from_numpy(np.ascontiguousarray(block)).to(self._device, non_blocking=False)
        else:
            if block.dim() != 2 or block.shape[0] != self._n_chans:
                raise ValueError("`block` must have shape [n_chans, times].")
            t = block.to(self._device)
        t = t.to(self._t_in)
        need = self._held + t.shape[1]
        if self._buf is None or self._buf.shape[1] < need:
            grown = torch.empty((self._n_chans, max(need, 2 * (self._held + self._hw) + t.shape[1])),
                                dtype=self._t_in, device=self._device)
            if self._held:
                grown[:, : self._held] = self._buf[:, : self._held]
            self._buf = grown
        self._buf[:, self._held:need] = t
        self._held = need
        self._received += t.shape[1]

    def _emit(self, upto: int, n_total: int, as_numpy: bool):
        """Outputs [emitted, upto) of a recording that (as far as these outputs can tell) has
        ``n_total`` samples; afterwards drop the samples no later output can reach."""
        torch = _hip.require_gpu()
        count = upto - self._emitted
        if count <= 0:
            out = torch.empty((self._n_chans, 0), dtype=self._t_out, device=self._device)
        else:
            window = self._buf[:, : self._held]
            out = torch.empty((self._n_chans, count), dtype=self._t_out, device=self._device)
            self._plan.apply_window(window, self._first, self._emitted, count, n_total, out=out)
            self._emitted = upto
            keep_from = max(self._emitted - self._hw, self._first)  # the window form wants a symmetric halo
            drop = keep_from - self._first
            if drop > 0:
                remaining = self._held - drop
                self._buf[:, :remaining] = self._buf[:, drop:self._held].clone()
                self._first, self._held = keep_from, remaining
        return _hip.to_host_numpy(out) if as_numpy else out

    def push(self, block):
        """Feed the next ``[n_chans, k]`` block (NumPy array or CUDA tensor); returns the outputs that
        became computable, ``[n_chans, m]`` in the same container kind (``m`` may be 0)."""
        if self._closed:
            raise ValueError("This stream has been finished.")
        self._append(block)
        # output n is final once every later sample it reaches has arrived
        upto = max(self._received - self._ahead, self._emitted)
        return self._emit(upto, self._received, isinstance(block, np.ndarray))

    def finish(self, as_numpy: bool = True):
        """End of the recording: returns the remaining ``latency`` outputs, whose divisors count
        only the taps inside the recording (parrm.py:862-866), and closes the stream."""
        if self._closed:
            raise ValueError("This stream has been finished.")
        self._closed = True
        return self._emit(self._received, self._received, as_numpy)


def filter_file(filt, src_path, dst_path, chunk_samples: int = 0, out_dtype=None, device=None):
    """Filter a ``[channels, times]`` ``.npy`` recording on disk into a new ``.npy`` file.

    ``filt`` is the dense filter (``PARRM.filter``).  Both files are memory-mapped; time chunks of
    ``chunk_samples`` (default: ~256 MiB of input) travel mapped file -> page-locked buffer -> device ->
    ``parrm_filter_apply_window`` -> page-locked buffer -> mapped file with two chunks in flight.
    float32 recordings produce float64 output (the reference's rule) unless ``out_dtype=np.float32``.
    Returns the output's ``numpy.memmap`` (flushed)."""
    torch = _hip.require_gpu()
    hw, _, _ = _tap_reach(filt)
    src = np.load(src_path, mmap_mode="r")
    if src.ndim != 2:
        raise ValueError("`data` must be a 2D array.")
    if src.dtype not in (np.float32, np.float64):
        raise TypeError("the recording must be stored as float32 or float64.")
    out_dtype = np.dtype(np.float64 if out_dtype is None else out_dtype)
    if out_dtype == np.float32 and src.dtype != np.float32:
        raise TypeError("float32 output needs a float32 recording.")
    n_chans, n_samples = src.shape
    dst = np.lib.format.open_memmap(dst_path, mode="w+", dtype=out_dtype, shape=(n_chans, n_samples))
    if n_chans == 0 or n_samples == 0:
        dst.flush()
        return dst
    index = torch.cuda.current_device() if device is None else int(device)
    dev = torch.device("cuda", index)
    plan = _hip.FilterPlan(filt, device=index)
    if chunk_samples <= 0:
        chunk_samples = max((256 << 20) // (src.dtype.itemsize * n_chans), 16 * hw + 1024)
    chunk_samples = min(chunk_samples, n_samples)
    cap = chunk_samples + 2 * hw
    t_in = torch.float64 if src.dtype == np.float64 else torch.float32
    t_out = torch.float64 if out_dtype == np.float64 else torch.float32
    with torch.cuda.device(dev):
        lanes = []
        for _ in range(2):
            lanes.append({
                "h_in": torch.empty((n_chans, cap), dtype=t_in).pin_memory(),
                "h_out": torch.empty((n_chans, chunk_samples), dtype=t_out).pin_memory(),
                "d_in": torch.empty((n_chans, cap), dtype=t_in, device=dev),
                "d_out": torch.empty((n_chans, chunk_samples), dtype=t_out, device=dev),
                "stream": torch.cuda.Stream(device=dev),
                "done": None,   # (event, out_first, out_len) of the chunk in flight on this lane
            })

        def retire(lane):
            if lane["done"] is None:
                return
            event, o0, olen = lane["done"]
            event.synchronize()
            dst[:, o0:o0 + olen] = lane["h_out"][:, :olen].numpy()
            lane["done"] = None

        for k, o0 in enumerate(range(0, n_samples, chunk_samples)):
            lane = lanes[k % 2]
            retire(lane)  # chunk k-2: its copies are done, its buffers are free again
            olen = min(chunk_samples, n_samples - o0)
            b0, b1 = max(o0 - hw, 0), min(o0 + olen + hw, n_samples)
            blen = b1 - b0
            np.copyto(lane["h_in"][:, :blen].numpy(), src[:, b0:b1])
            with torch.cuda.stream(lane["stream"]):
                lane["d_in"][:, :blen].copy_(lane["h_in"][:, :blen], non_blocking=True)
                plan.apply_window(lane["d_in"][:, :blen], b0, o0, olen, n_samples, out=lane["d_out"][:, :olen])
                lane["h_out"][:, :olen].copy_(lane["d_out"][:, :olen], non_blocking=True)
                event = torch.cuda.Event()
                event.record()
            lane["done"] = (event, o0, olen)
        for lane in lanes:
            retire(lane)
    dst.flush()
    return dst
