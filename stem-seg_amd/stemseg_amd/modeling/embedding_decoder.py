"""Embedding head: 3-D squeeze-expand decoder -> per-voxel embedding | variance | (seediness) channels.

Drop-in for the reference's ``stemseg.modeling.embedding_decoder`` (class ``SqueezingExpandDecoder``
registered as "squeeze_expand_decoder" in ``EMBEDDING_HEAD_REGISTRY``, embedding_decoder.py:8-145):
same constructor signature (model_builder.py:284-291), same attributes read by the caller
(``embedding_size``, ``variance_channels``, ``seediness_channels``; inference_model.py:141-145), same
state-dict keys, same ``forward(list of 4 [N,C,T,h,w]) -> [N, E+Ev(+1), T, H/4, W/4]``.
The body is one call into the HIP decoder (libstemseg_hip.so) per sample.
"""
import torch
import torch.nn as nn

from ..utils.global_registry import GlobalRegistry
from .decoder_base import SqueezeExpandTrunk
from .embedding_utils import get_nb_embedding_dims, get_nb_free_dims, grid_axes, grid_vectors

EMBEDDING_HEAD_REGISTRY = GlobalRegistry.get("EmbeddingHead")

ACT_NONE, ACT_TANH_GRID, ACT_SIGMOID, ACT_BANDWIDTH, ACT_GRID = 0, 1, 2, 3, 4


@EMBEDDING_HEAD_REGISTRY.add("squeeze_expand_decoder")
class SqueezingExpandDecoder(SqueezeExpandTrunk):
    def __init__(self, in_channels, inter_channels, embedding_size, tanh_activation, seediness_output, experimental_dims,
                 ConvType=nn.Conv3d, PoolType=nn.AvgPool3d, NormType=nn.Identity, num_frames=None):
        if ConvType is not nn.Conv3d:
            raise NotImplementedError("HIP decoder implements nn.Conv3d stages only")
        super().__init__(in_channels, inter_channels, PoolType, NormType, num_frames)
        self.embedding_size = embedding_size
        self.variance_channels = embedding_size - get_nb_free_dims(experimental_dims)
        self.embedding_dim_mode = experimental_dims
        n_emb = get_nb_embedding_dims(experimental_dims)
        c4 = inter_channels[-1]
        self.conv_embedding = nn.Conv3d(c4, n_emb, kernel_size=1, padding=0, bias=False)
        self.conv_variance = nn.Conv3d(c4, self.variance_channels, kernel_size=1, padding=0, bias=True)
        self.conv_seediness, self.seediness_channels = None, 0
        if seediness_output:
            self.conv_seediness = nn.Conv3d(c4, 1, kernel_size=1, padding=0, bias=False)
            self.seediness_channels = 1
        n_head = n_emb + self.variance_channels + self.seediness_channels
        if n_head > 10:
            raise NotImplementedError("embedding head with %d output channels (mode '%s', EMBEDDING_SIZE %d%s): the fused HIP "
                                      "heads kernel emits at most 10 (the widest mode embedding_utils admits, xytff + seediness, needs 9)"
                                      % (n_head, experimental_dims, embedding_size, ", seediness" if seediness_output else ""))
        self.tanh_activation = tanh_activation
        self.register_buffer("time_scale", torch.tensor(1.0, dtype=torch.float32))
        # inference_model.py:148 applies exp()*10 to the variance channels afterwards; the pipeline can ask the
        # heads kernel to do it in the same pass instead (then the caller must NOT apply it again).
        self.fuse_bandwidth_activation = False

    def _head_spec(self):
        c4 = self.inter_channels[-1]
        ws = [self.conv_embedding.weight.reshape(-1, c4), self.conv_variance.weight.reshape(-1, c4)]
        bs = [torch.zeros(ws[0].shape[0], device=ws[0].device), self.conv_variance.bias]
        axes = grid_axes(self.embedding_dim_mode)
        if self.tanh_activation:
            act = [ACT_TANH_GRID] * len(axes)
        else:
            act = [ACT_GRID if a else ACT_NONE for a in axes]
        act += [ACT_NONE] * self.variance_channels
        axes = axes + [0] * self.variance_channels
        if self.conv_seediness is not None:
            ws.append(self.conv_seediness.weight.reshape(-1, c4))
            bs.append(torch.zeros(1, device=ws[0].device))
            act.append(ACT_SIGMOID)
            axes.append(0)
        return self._fold(torch.cat(ws, 0)), torch.cat([b.float() for b in bs], 0), act, axes

    def _grid(self, c, T, H4, W4, dev):
        # the time_scale buffer lives on the device: read it back once per value, not per call (a per-call .item() is a
        # host sync and is illegal inside hipGraph capture)
        sig = (self.time_scale.data_ptr(), self.time_scale._version)
        if c.get("ts_sig") != sig:
            c["ts_sig"], c["ts"] = sig, float(self.time_scale)
        key = (T, H4, W4, dev.index, c["ts"])
        if key not in c["grids"]:
            c["grids"][key] = tuple(g.contiguous() for g in grid_vectors(H4, W4, T, c["ts"], device=dev))
        return c["grids"][key]

    def _acts(self):
        act = list(self._packed()["act"])
        if self.fuse_bandwidth_activation:
            n_emb = get_nb_embedding_dims(self.embedding_dim_mode)
            for o in range(n_emb, n_emb + self.variance_channels):
                act[o] = ACT_BANDWIDTH
        return act

    @torch.no_grad()
    def forward(self, x):
        """x: list of 4 feature stacks [N, C, T, h, w] ordered 32x, 16x, 8x, 4x  ->  [N, E+Ev(+1), T, H/4, W/4]"""
        assert len(x) == 4, "Expected 4 feature maps, got {}".format(len(x))
        act = self._acts()
        return torch.stack([self.run_hip([f[n] for f in x], 0, act) for n in range(x[0].shape[0])], 0)

    @torch.no_grad()
    def forward_single(self, feats, input_layout, clip_batch=None):
        """One sample, encoder-native layouts (1: [T,C,h,w]; 2: (zero-haloed buffers, (T,H4,W4))); ``clip_batch``: see run_hip."""
        return self.run_hip(feats, input_layout, self._acts(), clip_batch=clip_batch)
