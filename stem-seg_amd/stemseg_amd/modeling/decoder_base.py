"""Shared machinery of the two squeeze-expand decoders: reference-compatible parameter containers
(state-dict keys of SURVEY.md section 5) + the call into ``stemseg_hip_decoder_forward``.

The ``nn.Conv3d`` / ``nn.GroupNorm`` objects created here are *parameter holders only* -- they give
checkpoints the reference's key names (``block_32x.0.weight`` ... ``conv_4.weight``); ``forward`` never
calls them.  All arithmetic happens in libstemseg_hip.so.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import hip
from ..config import cfg
from .common import POOL_TABLE, TSCALE_TABLE

_BLOCK_CONVS = (("block_32x", 0), ("block_32x", 4), ("block_32x", 8), ("block_16x", 0), ("block_16x", 4),
                ("block_8x", 0), ("block_4x", 0))


def _stage(cin, cout, NormType, pool_creator):
    # Conv3d(3, p=1) -> Norm -> ReLU -> Pool|Identity   (indices 0,1,2,3 within the block)
    return [nn.Conv3d(cin, cout, 3, stride=1, padding=1), NormType(cout), nn.ReLU(inplace=True),
            pool_creator(3, stride=(2, 1, 1), padding=1) if pool_creator else None]


class SqueezeExpandTrunk(nn.Module):
    """Parameter layout + HIP execution shared by embedding_decoder / seediness_decoder."""

    def __init__(self, in_channels, inter_channels, PoolType=nn.AvgPool3d, NormType=nn.Identity, num_frames=None):
        super().__init__()
        assert len(inter_channels) == 4
        self.num_frames = cfg.INPUT.NUM_FRAMES if num_frames is None else num_frames
        if self.num_frames not in POOL_TABLE:
            raise NotImplementedError("clip length %r" % (self.num_frames,))
        if PoolType not in (nn.AvgPool3d, nn.MaxPool3d):      # the two entries of POOL_TYPE (model_builder.py:29-33)
            raise NotImplementedError("HIP decoder implements POOL_TYPE 'avg' (AvgPool3d) and 'max' (MaxPool3d)")
        self.pool_code = 2 if PoolType is nn.MaxPool3d else 1
        self.pool_flags, self.t_scales = POOL_TABLE[self.num_frames], TSCALE_TABLE[self.num_frames]
        probe = NormType(inter_channels[0])
        if isinstance(probe, nn.GroupNorm):
            self.gn_groups, self.gn_eps = probe.num_groups, probe.eps
        elif isinstance(probe, nn.Identity):                 # NORMALIZATION_LAYER 'none'
            self.gn_groups, self.gn_eps = 0, 0.0
        else:
            raise NotImplementedError("HIP decoder implements NORMALIZATION_LAYER 'gn' (GroupNorm) and 'none' (Identity)")
        self.in_channels, self.inter_channels = in_channels, list(inter_channels)

        def pools(n):
            return [(PoolType if on else (lambda *a, **k: nn.Identity())) for on in self.pool_flags[:n]]
        c32, c16, c8, c4 = inter_channels

        def block(cins, cout, poolers):
            mods = []
            for ci, pc in zip(cins, poolers):
                mods += [m for m in _stage(ci, cout, NormType, pc) if m is not None]
            return nn.Sequential(*mods)
        self.block_32x = block([in_channels, c32, c32], c32, pools(3))
        self.block_16x = block([in_channels, c16], c16, pools(2))
        self.block_8x = block([in_channels], c8, pools(1))
        self.block_4x = nn.Sequential(*[m for m in _stage(in_channels, c4, NormType, None) if m is not None])
        self.conv_16 = nn.Conv3d(c32 + c16, c16, 1, bias=False)
        self.conv_8 = nn.Conv3d(c16 + c8, c8, 1, bias=False)
        self.conv_4 = nn.Conv3d(c8 + c4, c4, 1, bias=False)
        self._cache = {}          # precision -> packed weights (valid while the parameter versions match)
        self.fold_conv4 = True    # conv_4 folded into the heads at load (see _fold; False: the two-step form, for A/B and exactness checks)
        self.fold_linear_tail = True     # the whole linear tail folded into per-level head matrices (see _linear_tail; narrow heads only)
        self._retired = []        # superseded packings, kept alive for captured graphs
        self._workspaces, self._ws_desc = {}, {}     # (T, H4, W4, layout, device, lane) -> workspace tensor / the descriptor it was sized for
        self.input_layout = 0     # 0: [C,T,h,w] per sample (reference API); 2: caller passes zero-haloed buffers
        self.concurrency = 1      # 0: single stream; k>=1: branch streams of the library's set k-1 (see stemseg_hip.h)
        self.detached = False     # True: the call does not join; ``join()`` must follow (twin-decoder overlap)
        self.precision = hip.DEFAULT_PRECISION    # hip.PRECISIONS: "f16x3" | "bf16x6" | "f32"
        self.lane = 0             # selects one of several independent workspaces (one per in-flight step / stream)

    # ---- to be provided by the concrete decoder ---------------------------------------------------
    def _head_spec(self):
        """-> (weight [n_out, c4] tensor, bias [n_out] tensor, act codes, grid_axis codes)"""
        raise NotImplementedError

    def _linear_tail(self, w_head):
        """Between the last GroupNorm + ReLU of every branch and the heads' activations the reference applies only linear maps (trilinear
        up-sampling, concatenation, the bias-free 1x1x1 convs conv_16 / conv_8 / conv_4, the 1x1x1 heads; embedding_decoder.py:64-80,112-143), and
        channel mixing commutes with up-sampling:  heads(x) = up(up(up(M32 x32) + M16 y16) + M8 y8) + M4 y4.  -> the four level matrices
        [n_out, c32], [n_out, c16], [n_out, c8], [n_out, c4] as fp64 products rounded once (StemsegDecoderWeights: fuse_w[0..2] = NULL,
        head_w = [M32 | M16 | M8 | M4])."""
        c32, c16, c8, c4 = self.inter_channels
        wh = w_head.detach().double()
        w4 = self.conv_4.weight.detach().reshape(c4, c8 + c4).double()
        w8 = self.conv_8.weight.detach().reshape(c8, c16 + c8).double()
        w16 = self.conv_16.weight.detach().reshape(c16, c32 + c16).double()
        m4 = wh @ w4[:, c8:]
        a = wh @ w4[:, :c8]
        m8 = a @ w8[:, c16:]
        b = a @ w8[:, :c16]
        m16 = b @ w16[:, c32:]
        m32 = b @ w16[:, :c32]
        return [m.float().contiguous() for m in (m32, m16, m8, m4)]

    def _fold(self, w_head):
        """conv_4 (1x1x1, no bias, no activation: embedding_decoder.py:80,129) feeds nothing but the 1x1x1 heads: with ``fold_conv4`` the head
        weights are multiplied by conv_4's at load (fp64 product, rounded once) and the decoder applies them to the last concat buffer directly
        (StemsegDecoderWeights.fuse_w[2] = NULL) -- one linear map for two, like FrozenBN folded into its convolution; the inter[3]-channel
        map is never computed, written or read.  w_head: dense [n_out, inter[3]] -> [n_out, inter[2] + inter[3]] (unchanged without the fold)."""
        if self.fold_linear_tail and w_head.shape[0] <= hip.MAX_HEAD_OUT:
            return torch.cat([m.reshape(-1) for m in self._linear_tail(w_head)])      # (flat [M32 | M16 | M8 | M4]: the narrow heads' form)
        if not self.fold_conv4:
            return w_head
        w4 = self.conv_4.weight.detach().reshape(self.conv_4.out_channels, -1)
        return (w_head.detach().double() @ w4.double()).float()

    # ---- weights --------------------------------------------------------------------------------------
    def _param_signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + (bool(self.fold_conv4), bool(self.fold_linear_tail))

    def _packed(self):
        # one packing per precision: an overflow re-run in bf16x6 (ClipPipeline.step_checked, GraphedStep.collect) must not throw the
        # f16x3 packing away and pack it again on the way back.  A packing superseded by a parameter update is RETIRED, not freed: a
        # captured hipGraph holds raw pointers into it (a replay then computes with the old weights -- re-capture after an update -- but
        # never reads freed memory)
        sig = self._param_signature()
        c = self._cache.setdefault(self.precision, {})
        if c.get("sig") != sig:
            dev = next(self.parameters()).device
            conv_w, conv_b, gn_w, gn_b = [], [], [], []
            for blk, idx in _BLOCK_CONVS:
                conv, gn = getattr(self, blk)[idx], getattr(self, blk)[idx + 1]
                conv_w.append(hip.pack_conv_weight_any(conv.weight.detach().float(), self.precision))
                conv_b.append(conv.bias.detach().float().contiguous())
                if self.gn_groups:
                    gn_w.append(gn.weight.detach().float().contiguous())
                    gn_b.append(gn.bias.detach().float().contiguous())
                else:                                        # no normalisation layer: scale 1, shift 0
                    gn_w.append(torch.ones(conv.out_channels, dtype=torch.float32, device=dev))
                    gn_b.append(torch.zeros(conv.out_channels, dtype=torch.float32, device=dev))
            hw, hb, act, axes = self._head_spec()
            lin = self.fold_linear_tail and len(act) <= hip.MAX_HEAD_OUT          # (the wide semseg head keeps conv_16 / conv_8 and folds conv_4 only)
            fuse = [None if lin else hip.pack_conv_weight_any(m.weight.detach().float(), self.precision) for m in (self.conv_16, self.conv_8)]
            fuse.append(None if (lin or self.fold_conv4) else hip.pack_conv_weight_any(self.conv_4.weight.detach().float(), self.precision))
            if c:
                self._retired.append(dict(c))
            c.clear()
            c.update(sig=sig, conv_w=conv_w, conv_b=conv_b, gn_w=gn_w, gn_b=gn_b, fuse=fuse,
                     head_w=hw.detach().float().contiguous().to(dev), head_b=hb.detach().float().contiguous().to(dev),
                     act=list(act), axes=list(axes), grids={})
        return c

    def _desc(self, T, H4, W4, layout, act):
        d = hip.DecoderDesc()
        d.struct_bytes = C.sizeof(hip.DecoderDesc)
        d.in_channels = self.in_channels
        for i in range(4):
            d.inter[i] = self.inter_channels[i]
        d.T, d.H4, d.W4 = T, H4, W4
        d.gn_groups, d.gn_eps = self.gn_groups, self.gn_eps
        for i in range(3):
            d.pool[i], d.t_scale[i] = self.pool_flags[i] * self.pool_code, self.t_scales[i]
        d.n_out = len(act)
        return d

    def _grid(self, c, T, H4, W4, dev):
        return None, None, None

    def check_workspaces(self):
        """Debug check (synchronises): clobbered guard words over all cached workspaces (stemseg_hip_decoder_check_workspace)."""
        bad, where = 0, []
        for key, ws in self._workspaces.items():
            n, first = C.c_int32(0), C.c_int64(-1)
            with torch.cuda.device(ws.device):
                hip.check(hip.lib().stemseg_hip_decoder_check_workspace(C.byref(self._ws_desc[key]), hip.ptr(ws), ws.numel(), C.byref(n), C.byref(first), hip.stream()))
            if n.value:
                bad += n.value
                where.append((key, first.value))
        return bad, where

    def join(self):
        """Make the current stream wait for a detached forward of this decoder."""
        hip.check(hip.lib().stemseg_hip_decoder_join(int(self.concurrency), hip.stream()))

    @torch.no_grad()
    def run_hip(self, feats, input_layout=None, act_override=None, clip_batch=None):
        """feats: 4 device tensors (32x,16x,8x,4x) for ONE sample: dense [C,T,h,w] (layout 0), dense [T,C,h,w]
        (layout 1) or zero-haloed flat buffers (layout 2, then T/H4/W4 must be given via ``feats_shape``).
        Returns [n_out, T, H4, W4].
        ``clip_batch = (n, strides)``: layout 2 only -- ``feats`` are clip 0's buffers and clip c's lie ``strides[level] * c`` floats
        further (hip.alloc_padded_batch); all n clips go through ONE launch per decoder stage (StemsegDecoderDesc.n_clips) and the
        result is [n, n_out, T, H4, W4].  A clip's output is bit-identical to its single-clip call."""
        hip.require_gpu()
        layout = self.input_layout if input_layout is None else input_layout
        nb = 1
        if clip_batch is not None:
            nb, bstrides = int(clip_batch[0]), [int(v) for v in clip_batch[1]]
            if layout != 2 or nb < 1 or len(bstrides) != 4:
                raise ValueError("clip_batch needs zero-haloed inputs (layout 2) and one stride per level")
        if layout == 2:
            bufs, (T, H4, W4) = feats
        else:
            bufs = [f.contiguous().float() for f in feats]
            f4 = bufs[3]
            T, H4, W4 = (f4.shape[1], f4.shape[2], f4.shape[3]) if layout == 0 else (f4.shape[0], f4.shape[2], f4.shape[3])
        if T != self.num_frames:
            raise ValueError("decoder built for %d-frame clips, got %d" % (self.num_frames, T))
        dev = bufs[0].device
        c = self._packed()
        act = list(c["act"]) if act_override is None else list(act_override)
        d = self._desc(T, H4, W4, layout, act)
        for o in range(d.n_out if d.n_out <= 10 else 0):     # wide linear heads (n_out % 32 == 0) carry no activation table
            d.act[o], d.grid_axis[o] = act[o], c["axes"][o]
        d.input_layout = layout
        d.concurrency = int(self.concurrency)
        d.detached = int(bool(self.detached) and self.concurrency >= 1)
        d.precision = hip.PRECISIONS[self.precision]
        if nb > 1:
            d.n_clips = nb
            for i in range(4):
                d.feat_clip_stride[i] = bstrides[i]
        key = (T, H4, W4, layout, dev.index, self.lane, nb)
        ws = self._workspaces.get(key)
        if ws is None:
            nbytes = hip.lib().stemseg_hip_decoder_workspace_bytes(C.byref(d))
            if nbytes == 0:
                raise RuntimeError("decoder: " + hip.lib().stemseg_hip_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            hip.check(hip.lib().stemseg_hip_decoder_init_workspace(C.byref(d), hip.ptr(ws), nbytes, hip.stream()))
            self._workspaces[key] = ws
            self._ws_desc[key] = d
        w = hip.DecoderWeights()
        for i in range(7):
            w.conv_w[i], w.conv_b[i] = c["conv_w"][i].data_ptr(), c["conv_b"][i].data_ptr()
            w.gn_w[i], w.gn_b[i] = c["gn_w"][i].data_ptr(), c["gn_b"][i].data_ptr()
        for i in range(3):
            w.fuse_w[i] = c["fuse"][i].data_ptr() if c["fuse"][i] is not None else None      # (conv_4 folded into the heads: NULL)
        w.head_w, w.head_b = c["head_w"].data_ptr(), c["head_b"].data_ptr()
        gt, gy, gx = self._grid(c, T, H4, W4, dev)
        if gt is not None:
            w.grid_t, w.grid_y, w.grid_x = gt.data_ptr(), gy.data_ptr(), gx.data_ptr()
        out = torch.empty((nb, d.n_out, T, H4, W4) if clip_batch is not None else (d.n_out, T, H4, W4), dtype=torch.float32, device=dev)
        fp = (C.c_void_p * 4)(*[b.data_ptr() for b in bufs])
        hip.check(hip.lib().stemseg_hip_decoder_forward(C.byref(d), C.byref(w), fp, hip.ptr(out), hip.ptr(ws), ws.numel(), hip.stream()))
        return out
