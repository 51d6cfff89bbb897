"""Clip-length topology tables and the trilinear up-sampling module.

Mirror of stemseg/modeling/common.py: get_pooling_layer_creator :8-24, get_temporal_scales :27-35,
UpsampleTrilinear3D :69-78 (F.interpolate(mode='trilinear') -> hand-written HIP kernel).
"""
import torch
import torch.nn as nn

from .. import hip
from ..config import cfg

# NUM_FRAMES -> which of the (up to) three temporal pooling layers exist, and the temporal up-sampling factors
POOL_TABLE = {2: (0, 0, 0), 4: (1, 0, 0), 8: (1, 1, 0), 16: (1, 1, 1), 24: (1, 1, 1), 32: (1, 1, 1)}
TSCALE_TABLE = {2: (1, 1, 1), 4: (1, 1, 2), 8: (1, 2, 2), 16: (2, 2, 2), 24: (2, 2, 2), 32: (2, 2, 2)}


def get_pooling_layer_creator(PoolType, num_frames=None):
    n = cfg.INPUT.NUM_FRAMES if num_frames is None else num_frames
    if n not in POOL_TABLE:
        raise NotImplementedError()
    return [(lambda *a, **k: PoolType(*a, **k)) if on else (lambda *a, **k: nn.Identity(*a, **k)) for on in POOL_TABLE[n]]


def get_temporal_scales(num_frames=None):
    n = cfg.INPUT.NUM_FRAMES if num_frames is None else num_frames
    return list(TSCALE_TABLE[n]) if n in TSCALE_TABLE else None


class UpsampleTrilinear3D(nn.Module):
    """[N, C, T, H, W] -> scaled by integer ``scale_factor`` (t, h, w), align_corners=False, on the HIP kernel."""

    def __init__(self, size=None, scale_factor=None, align_corners=None):
        super().__init__()
        if size is not None or align_corners:
            raise NotImplementedError("HIP trilinear kernel: integer scale_factor, align_corners=False only")
        self.size, self.scale_factor, self.align_corners = size, scale_factor, align_corners

    @torch.no_grad()
    def forward(self, x):
        hip.require_gpu()
        sf = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor,) * 3
        st, sy, sx = (int(s) for s in sf)
        assert (st, sy, sx) == tuple(sf), "integer scale factors only"
        x = x.contiguous().float()
        return torch.stack([hip.upsample_trilinear(xi, st, sy, sx) for xi in x], 0)
