"""The sequence driver's hot-path half under the reference's module path (inference/main.py:23-49,52-170)."""
from stemseg_amd.inference.main import TrackGenerator, fg_masks_from_seediness, get_subsequence_frames  # noqa: F401
