"""MI355X-native PARRM engine: drop-in for the hot path of PyPARRM's ``PARRM`` class
(``find_period`` -> ``create_filter`` -> ``filter_data``), running on hand-written HIP kernels
for gfx950 behind the C ABI of ``include/parrm_hip.h``."""

__version__ = "0.2.0"

from .data import get_example_data_paths
from .parrm import PARRM, find_period_batched

__all__ = ["PARRM", "find_period_batched", "get_example_data_paths", "__version__"]
