"""``from stemseg.utils import Timer, RepoPaths`` (utils/__init__.py:1-3)."""
from stemseg_amd.utils.paths import RepoPaths  # noqa: F401
from stemseg_amd.utils.timer import Timer  # noqa: F401
