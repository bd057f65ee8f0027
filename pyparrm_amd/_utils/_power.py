"""Power spectra for QA plots / the parameter explorer, on the device.

Mirrors ``pyparrm._utils._power.compute_psd`` (reference ``src/pyparrm/_utils/_power.py:10-68``):
float32 periodogram of the FIRST ``n_points`` samples (``scipy.fft.fft(x, n)`` truncates or
zero-pads, :60), positive frequencies only, zero frequency dropped (:55, :60-62), scaled by
``1 / (fs * n_points)`` (:63-65).  The transform is rocFFT's real-to-complex FFT (through
``torch.fft.rfft``; the explorer calls this on every widget event with a few thousand points);
scaling and the cut at ``max_freq`` are fused element-wise device ops.

One reference accident is kept because results must match: ``psd[:-1] *= 2`` (:66) indexes the
CHANNEL axis of the ``[channels, frequencies]`` array, so every channel but the last is doubled
(the intent was every frequency but the Nyquist bin).
"""

from __future__ import annotations

import numpy as np

from .. import _hip


def compute_psd(data, sampling_freq, n_points: int, max_freq=None, n_jobs: int = 1):
    """Power spectral density of ``data[channels, times]`` (NumPy array or CUDA tensor).

    Returns ``(freqs, psd)``: ``freqs`` float64 NumPy ``[frequencies]``; ``psd`` float32
    ``[channels, frequencies]`` -- NumPy for NumPy input, a CUDA tensor for CUDA input.  ``n_jobs`` is
    accepted for signature compatibility (the device transform needs no thread pool).  Like the
    reference, no input checks are performed."""
    torch = _hip.require_gpu()
    # frequencies exactly as the reference forms them (:55-58)
    freqs = np.abs(np.fft.fftfreq(n_points, 1.0 / sampling_freq)[1:(n_points // 2) + 1])
    if max_freq is None:
        max_freq = freqs[-1]
    last = int(np.argwhere(freqs <= max_freq)[-1][0])

    on_device = not isinstance(data, np.ndarray)
    if on_device:
        x = data[..., :n_points].to(torch.float32)
    else:
        x = torch.from_numpy(np.ascontiguousarray(data[..., :n_points]).astype(np.float32)).cuda()
    coeffs = torch.fft.rfft(x, n=n_points, dim=-1)[..., 1:(n_points // 2) + 1]  # rocFFT R2C
    psd = coeffs.abs().to(torch.float32).square_().mul_(1.0 / (sampling_freq * n_points))
    psd[:-1] *= 2  # reference accident kept: the channel axis (see the module docstring)
    psd = psd[..., : last + 1]
    return freqs[: last + 1], (psd if on_device else psd.cpu().numpy())
