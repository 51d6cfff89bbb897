"""Semantic-segmentation head: the third twin of the squeeze-expand trunk, emitting raw class logits (+ fg channel).

Drop-in for ``stemseg.modeling.semseg_decoder`` (SEMSEG_HEAD_REGISTRY["squeeze_expand_decoder"],
semseg_decoder.py:12-116): ctor ``(in_channels, num_classes, inter_channels, feature_scales, foreground_channel=False,
PoolType=, NormType=)`` (model_builder.py:331-337), ``forward(list of 4 [N,C,T,h,w] ordered 4x, 8x, 16x, 32x -- note the
REVERSED order, :93) -> [N, num_classes(+1), T, H/4, W/4]``, state-dict keys incl. ``conv_out.weight``.
The wide linear head runs on the 1x1x1 MFMA conv (output channels zero-padded to a multiple of 32 inside).
"""
import torch
import torch.nn as nn

from .. import hip
from ..utils.global_registry import GlobalRegistry
from .decoder_base import SqueezeExpandTrunk

SEMSEG_HEAD_REGISTRY = GlobalRegistry.get("SemsegHead")


@SEMSEG_HEAD_REGISTRY.add("squeeze_expand_decoder")
class SqueezeExpandDecoder(SqueezeExpandTrunk):
    def __init__(self, in_channels, num_classes, inter_channels, feature_scales, foreground_channel=False,
                 ConvType=nn.Conv3d, PoolType=nn.AvgPool3d, NormType=nn.Identity, num_frames=None):
        if ConvType is not nn.Conv3d:
            raise NotImplementedError("HIP decoder implements nn.Conv3d stages only")
        assert tuple(feature_scales) == (4, 8, 16, 32)
        super().__init__(in_channels, inter_channels, PoolType, NormType, num_frames)
        self.is_3d = True
        self.out_channels = num_classes + 1 if foreground_channel else num_classes
        self.conv_out = nn.Conv3d(inter_channels[3], self.out_channels, kernel_size=1, padding=0, bias=False)
        self.has_foreground_channel = foreground_channel

    def _head_spec(self):
        n = self.out_channels
        if n <= 8:                                   # narrow: fused heads kernel, identity activation
            w = self._fold(self.conv_out.weight.reshape(n, -1))
            return w, torch.zeros(n, device=w.device), [0] * n, [0] * n
        npad = (n + 31) // 32 * 32                   # wide: 1x1x1 MFMA conv on zero-padded output channels
        wf = self._fold(self.conv_out.weight.detach().float().reshape(n, -1))
        w = torch.zeros(npad, wf.shape[1], 1, 1, 1, dtype=torch.float32, device=self.conv_out.weight.device)
        w[:n] = wf.reshape(n, -1, 1, 1, 1)
        packed = hip.pack_conv_weight_any(w, self.precision)
        return packed, torch.zeros(npad, device=w.device), [0] * npad, [0] * npad

    @torch.no_grad()
    def forward(self, x):
        assert len(x) == 4, "Expected 4 feature maps, got {}".format(len(x))
        x = x[::-1]                                  # 4x,8x,16x,32x -> 32x,16x,8x,4x
        return torch.stack([self.run_hip([f[n] for f in x], 0)[:self.out_channels] for n in range(x[0].shape[0])], 0)

    @torch.no_grad()
    def forward_single(self, feats, input_layout, clip_batch=None):
        """feats in the trunk's order (32x, 16x, 8x, 4x)."""
        out = self.run_hip(feats, input_layout, None, clip_batch=clip_batch)
        return out[:self.out_channels] if clip_batch is None else out[:, :self.out_channels]
