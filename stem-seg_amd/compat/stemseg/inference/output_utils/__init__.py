"""``from stemseg.inference.output_utils import YoutubeVISOutputGenerator, DavisOutputGenerator, KittiMOTSOutputGenerator``
(inference/main.py:7)."""
from stemseg_amd.inference.output_utils.generators import (DavisOutputGenerator, KittiMOTSOutputGenerator,  # noqa: F401
                                                           YoutubeVISOutputGenerator)
