"""MI355X-native PARRM engine: drop-in for PyPARRM's ``PARRM`` hot path."""

__version__ = "0.1.0"

from .data import get_example_data_paths
