"""The C-ABI library loads and exports every symbol include/parrm_hip.h declares (no compute)."""

import ctypes
import os
import re

from pyparrm_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "parrm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(parrm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_hip.SYMBOLS) == declared


def test_library_exports_every_symbol():
    lib = _hip.lib()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.parrm_hip_abi_version() == 3
    assert lib.parrm_hip_shutdown() == 0  # nothing allocated yet: a no-op that must not need a GPU
    n = ctypes.c_int(-1)
    assert lib.parrm_hip_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_argument_errors_are_reported_without_a_gpu():
    lib = _hip.lib()
    info = _hip.PlanInfo()
    assert lib.parrm_filter_plan_query(None, ctypes.byref(info)) != 0
    assert b"NULL" in lib.parrm_hip_last_error()
    assert lib.parrm_fit_workspace_bytes(25001, 256, 1, 20) > 0
    assert lib.parrm_fit_workspace_bytes(25001, 256, 1, 99) == 0
    assert lib.parrm_absdiff_workspace_bytes(256, 10_000_000) >= 256 * 8
