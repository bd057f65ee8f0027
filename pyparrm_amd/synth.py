"""Synthetic DBS-like recordings (the generator SURVEY.md section 8d defines; the reference
ships none -- its tests use white noise, tests/test_parrm.py:30-31).

x[c, n] = s[c, n] + g_c * A((n + phi_c) mod T*),  T* = fs / f_art * (1 + detune)
  s    unit-variance Gaussian background, per-channel stream
  A    20 harmonics of the period with 1/k amplitudes and fixed phases
  g_c  in [5, 20], phi_c in [0, T*)
"""

from __future__ import annotations

import numpy as np

N_HARMONICS = 20


def _artefact_params(n_chans: int, period: float):
    phases = np.random.default_rng(7).uniform(0.0, 2.0 * np.pi, N_HARMONICS)
    r = np.random.default_rng(8)
    gains = r.uniform(5.0, 20.0, n_chans)
    offsets = r.uniform(0.0, period, n_chans)
    return phases, gains, offsets


def synth_recording(n_chans: int, n_samples: int, sampling_freq: float, artefact_freq: float,
                    seed: int = 0, detune: float = 3e-5, dtype=np.float64) -> np.ndarray:
    """Host (NumPy) recording, channel c seeded with ``1000 + seed + c``."""
    period = sampling_freq / artefact_freq * (1.0 + detune)
    phases, gains, offsets = _artefact_params(n_chans, period)
    n = np.arange(n_samples, dtype=np.float64)
    out = np.empty((n_chans, n_samples), dtype=dtype)
    for c in range(n_chans):
        theta = (2.0 * np.pi / period) * (n + offsets[c])
        art = np.zeros(n_samples)
        for k in range(1, N_HARMONICS + 1):
            art += np.sin(k * theta + phases[k - 1]) / k
        noise = np.random.default_rng(1000 + seed + c).standard_normal(n_samples)
        out[c] = noise + gains[c] * art
    return out


def synth_recording_device(n_chans: int, n_samples: int, sampling_freq: float, artefact_freq: float,
                           seed: int = 0, detune: float = 3e-5, dtype=None, device="cuda",
                           chunk: int = 1 << 22, chan_range=None):
    """Same distribution generated directly in HBM (torch RNG for the background; the artefact
    term is identical to :func:`synth_recording`).  Channel ``c`` is a function of ``(seed, c)`` only
    -- its own generator stream, gain and phase offset -- so ``chan_range=(lo, hi)`` yields exactly
    rows ``lo:hi`` of the ``n_chans``-channel recording: the ranks of a channel-sharded run build
    their blocks of ONE recording without ever holding the rest.  Built in time chunks to bound
    temporaries."""
    import torch

    dtype = dtype or torch.float64
    period = sampling_freq / artefact_freq * (1.0 + detune)
    phases, gains, offsets = _artefact_params(n_chans, period)
    lo_c, hi_c = (0, n_chans) if chan_range is None else chan_range
    rows = hi_c - lo_c
    gen = torch.Generator(device=device)
    out = torch.empty((rows, n_samples), dtype=dtype, device=device)
    for c in range(lo_c, hi_c):
        gen.manual_seed(1000 + seed + 7 * c)
        out[c - lo_c] = torch.randn(n_samples, dtype=torch.float64, device=device, generator=gen).to(dtype)
    d_ph = torch.from_numpy(phases).to(device)
    d_gain = torch.from_numpy(gains[lo_c:hi_c]).to(device)[:, None]
    d_off = torch.from_numpy(offsets[lo_c:hi_c]).to(device)[:, None]
    w0 = 2.0 * np.pi / period
    for lo in range(0, n_samples, chunk):
        hi = min(lo + chunk, n_samples)
        n = torch.arange(lo, hi, dtype=torch.float64, device=device)[None, :]
        theta = w0 * (n + d_off)
        art = torch.zeros((rows, hi - lo), dtype=torch.float64, device=device)
        for k in range(1, N_HARMONICS + 1):
            art += torch.sin(k * theta + d_ph[k - 1]) / k
        out[:, lo:hi] = (out[:, lo:hi].to(torch.float64) + d_gain * art).to(dtype)
        del art, theta
    return out


def pulse_shape(u: np.ndarray) -> np.ndarray:
    """Artefact waveform over one period, ``u`` in [0, 1): a biphasic stimulation pulse (two
    triangular lobes) on a smooth cubic swell.  Only +, -, *, abs and max are used -- every one
    exactly rounded -- so the samples are bit-identical on any IEEE-754 host (no libm ``sin``,
    whose last bit depends on the SIMD path NumPy picks for the CPU at hand)."""
    lobe_a = np.maximum(0.0, 1.0 - np.abs(u - 0.25) * 20.0)
    lobe_b = np.maximum(0.0, 1.0 - np.abs(u - 0.35) * 12.0)
    swell = (u * (1.0 - u)) * (u * (1.0 - u)) * (1.0 - 2.0 * u)
    return lobe_a - 0.5 * lobe_b + 6.0 * swell


def synth_recording_exact(n_chans: int, n_samples: int, period: float, seed: int = 0,
                          gain_range=(2.0, 10.0), dtype=np.float64) -> np.ndarray:
    """Recording for cross-host golden fixtures: Gaussian background (NumPy's ziggurat, seeded
    per channel with ``seed + c``) plus ``gain_c * pulse_shape(((n + off_c) / period) mod 1)``.
    Everything a fixture needs to store is ``(n_chans, n_samples, period, seed)``; generated in
    time chunks so that 10 M-sample channels do not need gigabytes of temporaries."""
    r = np.random.default_rng(seed + 7919)
    gains = r.uniform(gain_range[0], gain_range[1], n_chans)
    offsets = r.uniform(0.0, period, n_chans)
    out = np.empty((n_chans, n_samples), dtype=dtype)
    chunk = 1 << 21
    for c in range(n_chans):
        rng = np.random.default_rng(seed + c)
        for lo in range(0, n_samples, chunk):
            hi = min(lo + chunk, n_samples)
            n = np.arange(lo, hi, dtype=np.float64)
            u = np.mod((n + offsets[c]) / period, 1.0)
            out[c, lo:hi] = rng.standard_normal(hi - lo) + gains[c] * pulse_shape(u)
    return out
