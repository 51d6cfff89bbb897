"""``from stemseg.config import cfg`` (config/__init__.py:1) -> the hot path's own cfg (same key names, the subset it reads)."""
from stemseg_amd.config import cfg, load_preset  # noqa: F401
