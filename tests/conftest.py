"""pytest configuration: registers the ``gpu`` marker and shared paths/helpers."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# a native backtrace on stderr should the process abort inside a library (parrm_capi.hip, AbortTrace): read when the
# HIP library is loaded
os.environ.setdefault("PARRM_ABORT_TRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # keep the evidence if the process ever dies on a signal (round 1 lost the log of an exit-time
    # crash): the Python-level stacks of all threads go to gpurun_out/, which gpurun merges back
    import faulthandler

    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        config._parrm_fault_log = open(os.path.join(out_dir, f"faulthandler_{os.getpid()}.log"), "w")
        faulthandler.enable(file=config._parrm_fault_log, all_threads=True)
    except OSError:
        faulthandler.enable()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def pytest_sessionfinish(session, exitstatus):
    """Drain the device and drop what the tests left behind while the HIP runtime is certainly still
    up: objects that own device or page-locked memory (plans, workspaces, read-back arrays) are then
    released here and not during interpreter shutdown."""
    import gc

    gc.collect()
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
