"""``from stemseg.modeling.backbone import BACKBONE_REGISTRY`` (model_builder.py:9)."""
from stemseg_amd.modeling.backbone import BACKBONE_REGISTRY, build_resnet_fpn_backbone  # noqa: F401
