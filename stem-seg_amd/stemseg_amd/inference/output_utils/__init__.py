"""Mask materialisation for the per-dataset writers.

Counterpart of the device-side half of ``stemseg/inference/output_utils/{davis,youtube_vis,kitti_mots}.py``: choose the
instances to keep, turn the stitched per-point labels into full-resolution masks.  The file formats (PNG palette,
COCO-RLE json, MOTS txt) stay with the caller.
"""
from .masks import MaskMaterializer, instances_to_keep  # noqa: F401
