"""``from stemseg.data import DavisUnsupervisedPaths, YoutubeVISPaths, KITTIMOTSPaths`` (inference/main.py:12)."""
from stemseg_amd.utils.paths import DavisUnsupervisedPaths, KITTIMOTSPaths, YoutubeVISPaths  # noqa: F401
from .generic_video_dataset_parser import parse_generic_video_dataset  # noqa: F401
