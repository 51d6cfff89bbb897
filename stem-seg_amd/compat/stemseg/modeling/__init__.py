"""Parent package of the hot-path modules that stemseg_amd.overlay.HOT_PATH serves (modeling/__init__.py:1 exports build_model)."""
from stemseg.modeling.model_builder import build_model  # noqa: F401
