"""``build_model`` for inference: backbone + heads from the global cfg, without the training side.

Counterpart of ``stemseg/modeling/model_builder.py``: ``build_model`` :247-369 (registry look-ups by the cfg's type strings,
constructor contracts of SURVEY.md section 8(b)) and the inference-time surface of ``TrainingModel`` :37-73,154-169 that
``modeling/inference_model.py`` uses (``backbone``, the three heads, their feature-map scale lists, ``run_backbone``).
Losses, the training ``forward`` and mask resizing (:101-153,171-245) are out of scope.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import config as _config
from ..config import cfg
from .backbone import BACKBONE_REGISTRY
from .embedding_decoder import EMBEDDING_HEAD_REGISTRY
from .seediness_decoder import SEEDINESS_HEAD_REGISTRY
from .semseg_decoder import SEMSEG_HEAD_REGISTRY


class InferenceOnlyModel(nn.Module):
    """State-dict compatible with the reference's TrainingModel (keys ``backbone.*``, ``embedding_head.*``,
    ``seediness_head.*``, ``semseg_head.*``)."""

    def __init__(self):
        super().__init__()
        self.backbone = None
        self.embedding_head = self.seediness_head = self.semseg_head = None
        self.embedding_head_feature_map_scale = self.seediness_head_feature_map_scale = self.semseg_feature_map_scale = None

    embedding_head_output_scale = property(lambda self: min(self.embedding_head_feature_map_scale))
    semseg_output_scale = property(lambda self: min(self.semseg_feature_map_scale) if self.semseg_feature_map_scale else 4)
    feature_map_scales = (4, 8, 16, 32)

    @torch.no_grad()
    def run_backbone(self, image_seqs):
        """ImageList-like (``.tensors`` [N,T,3,H,W]) or a float tensor [..., 3, H, W] -> OrderedDict {4, 8, 16, 32: [N*T, 256,
        H/s, W/s]} (model_builder.py:154-169)."""
        x = image_seqs.tensors if hasattr(image_seqs, "tensors") else image_seqs
        x = x.reshape((-1,) + tuple(x.shape[-3:])).contiguous().float()
        return OrderedDict(zip(self.feature_map_scales, self.backbone(x)))

    def forward(self, image_seqs, targets):
        raise NotImplementedError("training is outside the MI355X hot path (SURVEY.md section 2); use InferenceModel")


# model_builder.py:29-33 (POOLER_REGISTRY / NORM_REGISTRY of the reference)
_POOLERS = {"avg": nn.AvgPool3d, "max": nn.MaxPool3d}


def _norm(kind, groups):
    if kind == "gn":
        return lambda c: nn.GroupNorm(groups, c)
    if kind == "none":
        return lambda c: nn.Identity()
    raise ValueError("NORMALIZATION_LAYER '%s' (gn | none)" % kind)


def build_model(restore_pretrained_backbone_wts=False, logger=None):
    """backbone + heads from the global cfg (model_builder.py:247-369 minus losses / pretrained-weight restore)."""
    _config.refresh()
    if restore_pretrained_backbone_wts:
        raise NotImplementedError("pretrained-backbone restore belongs to training (model_builder.py:259-277)")
    m = InferenceOnlyModel()
    m.backbone = BACKBONE_REGISTRY[cfg.MODEL.BACKBONE.TYPE](cfg)
    e = cfg.MODEL.EMBEDDINGS
    norm = _norm(e.NORMALIZATION_LAYER, e.GN_NUM_GROUPS)
    m.embedding_head = EMBEDDING_HEAD_REGISTRY[e.HEAD_TYPE](
        m.backbone.out_channels, e.INTER_CHANNELS, e.EMBEDDING_SIZE, tanh_activation=e.TANH_ACTIVATION,
        seediness_output=not cfg.MODEL.USE_SEEDINESS_HEAD, experimental_dims=cfg.MODEL.EMBEDDING_DIM_MODE,
        PoolType=_POOLERS[e.POOL_TYPE], NormType=norm)
    m.seediness_head = None
    if cfg.MODEL.USE_SEEDINESS_HEAD:
        s = cfg.MODEL.SEEDINESS
        m.seediness_head = SEEDINESS_HEAD_REGISTRY[s.HEAD_TYPE](
            m.backbone.out_channels, s.INTER_CHANNELS, PoolType=_POOLERS[s.POOL_TYPE], NormType=_norm(s.NORMALIZATION_LAYER, s.GN_NUM_GROUPS))
    m.semseg_head = None
    if cfg.MODEL.USE_SEMSEG_HEAD:
        g = cfg.MODEL.SEMSEG
        m.semseg_head = SEMSEG_HEAD_REGISTRY[g.HEAD_TYPE](
            m.backbone.out_channels, cfg.INPUT.NUM_CLASSES, inter_channels=g.INTER_CHANNELS, feature_scales=g.FEATURE_SCALE,
            foreground_channel=g.FOREGROUND_CHANNEL, PoolType=_POOLERS[g.POOL_TYPE], NormType=_norm(g.NORMALIZATION_LAYER, g.GN_NUM_GROUPS))
        m.semseg_feature_map_scale = list(g.FEATURE_SCALE)
    m.embedding_head_feature_map_scale = list(e.SCALE)
    m.seediness_head_feature_map_scale = list(cfg.MODEL.SEEDINESS.FEATURE_SCALE)
    return m
