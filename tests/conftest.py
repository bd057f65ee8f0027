"""pytest configuration: registers the ``gpu`` marker and shared paths/helpers."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
