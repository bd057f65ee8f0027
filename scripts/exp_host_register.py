#!/usr/bin/env python3
"""What hipHostRegister does with ranges that share a page (no GPU access to the memory is made)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyparrm_amd import _hip

_hip.require_gpu()
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipGetErrorName.restype = C.c_char_p
hip.hipGetErrorName.argtypes = [C.c_int]


def name(rc):
    return hip.hipGetErrorName(rc).decode()


raw = np.zeros(8 * 4096, dtype=np.uint8)
base = (raw.ctypes.data + 4095) // 4096 * 4096  # page-aligned start inside raw
print("base page", hex(base))
a, alen = base + 100, 6000            # pages 0..1
b, blen = base + 100 + 6000, 6000     # pages 1..2 (shares page 1 with a)
print("register a:", name(hip.hipHostRegister(a, alen, 0)))
print("register b (shares a page with a):", name(hip.hipHostRegister(b, blen, 0)))
print("register a again:", name(hip.hipHostRegister(a, alen, 0)))
print("register sub-range of a:", name(hip.hipHostRegister(a + 16, 64, 0)))
print("unregister a:", name(hip.hipHostUnregister(a)))
print("unregister b:", name(hip.hipHostUnregister(b)))
print("unregister a (second time):", name(hip.hipHostUnregister(a)))
hip.hipGetLastError()
# a pageable copy through torch, then an explicit registration of the same array, then release
x = np.ones(3_000_000, dtype=np.float64)
d = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
print("register array that went through a pageable H2D copy:", name(hip.hipHostRegister(x.ctypes.data, x.nbytes, 0)))
print("unregister it:", name(hip.hipHostUnregister(x.ctypes.data)))
back = d.cpu()
torch.cuda.synchronize()
print("pageable D2H after that ok:", bool((back == 1).all()))
d2 = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
print("pageable H2D after that ok:", bool((d2 == 1).all().item()))
