"""Helpers next to the hot path (mirrors the reference's ``pyparrm._utils`` package layout)."""
