"""Seediness head: the same 3-D squeeze-expand trunk with one sigmoid output channel.

Drop-in for ``stemseg.modeling.seediness_decoder`` (SEEDINESS_HEAD_REGISTRY["squeeze_expand_decoder"],
seediness_decoder.py:8-112): ctor ``(in_channels, inter_channels, PoolType=, NormType=)`` (model_builder.py:310-314),
``forward(list of 4 [N,C,T,h,w]) -> [N,1,T,H/4,W/4]``, state-dict keys incl. ``conv_out.weight``.
"""
import torch
import torch.nn as nn

from ..utils.global_registry import GlobalRegistry
from .decoder_base import SqueezeExpandTrunk

SEEDINESS_HEAD_REGISTRY = GlobalRegistry.get("SeedinessHead")


@SEEDINESS_HEAD_REGISTRY.add("squeeze_expand_decoder")
class SqueezingExpandDecoder(SqueezeExpandTrunk):
    def __init__(self, in_channels, inter_channels, ConvType=nn.Conv3d, PoolType=nn.AvgPool3d, NormType=nn.Identity, num_frames=None):
        if ConvType is not nn.Conv3d:
            raise NotImplementedError("HIP decoder implements nn.Conv3d stages only")
        super().__init__(in_channels, inter_channels, PoolType, NormType, num_frames)
        self.conv_out = nn.Conv3d(inter_channels[3], 1, kernel_size=1, padding=0, bias=False)

    def _head_spec(self):
        w = self._fold(self.conv_out.weight.reshape(1, -1))
        return w, torch.zeros(1, device=w.device), [2], [0]     # sigmoid, no grid

    @torch.no_grad()
    def forward(self, x):
        assert len(x) == 4
        return torch.stack([self.run_hip([f[n] for f in x], 0) for n in range(x[0].shape[0])], 0)

    @torch.no_grad()
    def forward_single(self, feats, input_layout, clip_batch=None):
        return self.run_hip(feats, input_layout, None, clip_batch=clip_batch)
