"""Generate + compile (hipRTC, no GPU needed) the per-filter comb kernel for a sampling/artefact frequency pair and
store the code object (and its source) in a directory -- by default the in-tree pyparrm_amd/lib/kernels/ that
the library searches first.  `__graft_entry__.build()` calls this for the BASELINE geometry."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_filter(period, n_samples=10_000_000):
    """The reference's default filter design (parrm.py:788-833) for an artefact period in samples."""
    phw = period / 50
    hw, hits = 0, 0
    while hits < 50 and hw < (n_samples - 1) // 2:
        hw += 1
        if np.mod(hw, period) <= phw:
            hits += 1
    w = np.arange(-hw, hw + 1)
    mask = ((np.mod(w, period) <= phw) | (np.mod(w, period) >= period - phw)) & (np.abs(w) > 0)
    f = -mask.astype(np.float64) / max(mask.sum(), 1)
    f[hw] = 1.0
    return f


def precompile(filt, out_dir, stride=0):
    lib = ctypes.CDLL(os.path.join(ROOT, "pyparrm_amd", "lib", "libparrm_hip.so"))
    lib.parrm_filter_comb_precompile.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_char_p,
                                                 ctypes.c_char_p, ctypes.c_size_t]
    lib.parrm_hip_last_error.restype = ctypes.c_char_p
    filt = np.ascontiguousarray(filt, dtype=np.float64)
    buf = ctypes.create_string_buffer(4096)
    rc = lib.parrm_filter_comb_precompile(filt.ctypes.data, filt.size, stride, out_dir.encode(), buf, 4096)
    if rc != 0:
        raise RuntimeError(lib.parrm_hip_last_error().decode())
    return buf.value.decode()


if __name__ == "__main__":
    fs, fa = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (22000.0, 130.0)
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "pyparrm_amd", "lib", "kernels")
    period = fs / fa * (1 + 3e-5)
    print(precompile(default_filter(period), out))
