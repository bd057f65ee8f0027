"""Locate the example recordings (mirrors ``pyparrm.data.example_data``,
src/pyparrm/data/example_data.py:13-30: same names, same error text)."""

from pathlib import Path

_DIR = Path(__file__).resolve().parent / "example_data"

DATASETS = {
    name: f"{name}.npy"
    for name in ("example_data", "example_data_artefact_free", "matlab_filtered", "ecog_lfp_data")
}


def get_example_data_paths(name: str) -> str:
    """Return the path of the example recording called ``name``."""
    if name not in DATASETS:
        raise ValueError(f"`name` must be one of: {list(DATASETS.keys())}")
    return str(_DIR / DATASETS[name])
