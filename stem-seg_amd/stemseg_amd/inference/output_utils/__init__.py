"""Mask materialisation for the per-dataset writers.

Counterpart of the device-side half of ``stemseg/inference/output_utils/{davis,youtube_vis,kitti_mots}.py``: choose the
instances to keep, turn the stitched per-point labels into full-resolution masks (``MaskMaterializer``); ``generators``
wraps it in the reference's output-generator call contract (DAVIS PNG writer complete, the RLE formats left to the caller).
"""
from .generators import DavisOutputGenerator, KittiMOTSOutputGenerator, YoutubeVISOutputGenerator  # noqa: F401
from .masks import MaskMaterializer, instances_to_keep  # noqa: F401
