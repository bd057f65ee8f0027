"""Example recordings shipped with the package (data files of the reference,
``src/pyparrm/data/example_data/*.npy``; used as golden fixtures)."""

from .example_data import DATASETS, get_example_data_paths

__all__ = ["DATASETS", "get_example_data_paths"]
