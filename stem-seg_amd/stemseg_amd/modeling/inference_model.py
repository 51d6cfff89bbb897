"""Sequence driver: frames -> (per-frame encoder, cached) -> per-clip head outputs, all resident on the GPU.

Counterpart of ``stemseg.modeling.inference_model.InferenceModel`` (:17-41 ctor, :63-194 forward) and of the
pre-processing helpers it relies on (data/inference_image_loader.py:23-43, data/common.py:12-30,142-159,
structures/image_list.py:58-107).  Same ``forward(images, subseq_idxes)`` contract and the same
``{"fg_masks", "multiclass_masks", "embeddings": [EmbeddingMapEntry(subseq_frames, embeddings, bandwidths, seediness)]}``
result, with three deliberate differences:
  * the clip's un-cached frames go through the encoder as one batch (reference: batch 1 per frame);
  * FPN features are written once into the zero-haloed layout both decoders consume (no torch.stack copy,
    no second pad for the seediness decoder);
  * outputs stay on the device unless ``outputs_on_cpu=True`` (reference: ``.cpu()`` per clip, then ``.cuda()``
    again in the chainer).
"""
import math
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from .. import config as _config
from ..config import cfg
from .embedding_utils import get_nb_free_dims  # noqa: F401  (re-export, as in the reference's import surface)
from .model_builder import build_model

EmbeddingMapEntry = namedtuple("EmbeddingMapEntry", ["subseq_frames", "embeddings", "bandwidths", "seediness"])


def compute_resize_params_2(image_dims, min_resize_dim, max_resize_dim):
    """(width, height) -> (new_width, new_height, scale): min side -> MIN_DIM unless the max side would exceed
    MAX_DIM; Python ``round`` (data/common.py:142-159)."""
    lower, higher = float(min(image_dims)), float(max(image_dims))
    scale = min_resize_dim / lower
    if higher * scale > max_resize_dim:
        scale = max_resize_dim / higher
    width, height = image_dims
    return round(scale * width), round(scale * height), scale


def pad_to_multiple_of_32(h, w):
    return int(math.ceil(h / 32)) * 32, int(math.ceil(w / 32)) * 32          # image_list.py:93-95


@torch.no_grad()
def preprocess_frames(frames, device="cuda"):
    """uint8 BGR frames [T,H0,W0,3] (numpy or tensor) -> (float32 [T,3,H,W] on the device, (new_h, new_w)): bilinear resize
    to cfg MIN/MAX_DIM, mean-subtract (no /255, std 1 with the reference's configs), zero-pad right/bottom to multiples of
    32 -- one HIP launch (inference_image_loader.py:23-43, data/common.py:12-30, image_list.py:93-104)."""
    hip.require_gpu()
    _config.refresh()
    x = torch.as_tensor(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] == 3, "frames: uint8 [T, H, W, 3]"
    x = x.to(device).contiguous()
    H0, W0 = x.shape[1:3]
    nw, nh, _ = compute_resize_params_2((W0, H0), cfg.INPUT.MIN_DIM, cfg.INPUT.MAX_DIM)
    H, W = pad_to_multiple_of_32(nh, nw)
    out = hip.preprocess_frames(x, (nh, nw), (H, W), cfg.INPUT.IMAGE_MEAN, cfg.INPUT.IMAGE_STD,
                                cfg.INPUT.NORMALIZE_TO_UNIT_SCALE, not cfg.INPUT.BGR_INPUT)
    return out, (nh, nw)


class InferenceModel(nn.Module):
    def __init__(self, restore_path=None, cpu_workers=0, preload_images=False, semseg_output_type=None, resize_scale=1.0,
                 semseg_generation_on_gpu=True, outputs_on_cpu=False):
        super().__init__()
        with torch.no_grad():
            self._model = build_model(restore_pretrained_backbone_wts=False)
        if restore_path:
            self.load_checkpoint_state(torch.load(restore_path, map_location="cpu")['model'])
        if float(resize_scale) < 1 or float(resize_scale) != int(resize_scale):
            raise NotImplementedError("resize_scale %r: the HIP trilinear kernel takes positive integer scales (the reference "
                                      "passes 1.0, or 4.0 under --resize_embeddings, inference/main.py:209-213)" % (resize_scale,))
        self.resize_scale = resize_scale
        self.semseg_output_type = semseg_output_type
        self.outputs_on_cpu = outputs_on_cpu
        self.EmbeddingMapEntry = EmbeddingMapEntry
        self._pads = {}
        self._pad_blocks = []
        self._last_written = {}          # (T, H, W, device, lane) -> the block the last encoder pass of that lane filled
        self.batch_decoders = True       # the clips of an encoder pass go through each decoder stage in one launch (False: clip by clip; A/B, same bits)
        self.overlap_decoders = True     # seediness decoder on a side stream + branch streams inside each decoder
        self.lane = 0
        self.eval()

    @property
    def plan_frames(self):
        """Frames every encoder launch is PLANNED for (ResNetFPN.plan_frames, default 32): one value per job and per set of ranks that
        must agree bit for bit.  A deployment that only ever passes lone 8-frame clips (the latency path) may set 8 -- its small-map
        layers then split K as a lone clip wants (about 0.5 % faster there) -- at the price of bits that differ from a 32-frame plan."""
        return self._model.backbone.plan_frames

    @plan_frames.setter
    def plan_frames(self, n):
        self._model.backbone.plan_frames = int(n)

    semseg_outputs_on_cpu = False       # True under stemseg_amd.overlay: the reference's writers index these on the host
    overflow_fallback = "bf16x6"        # mode a sequence is re-run in when a head output comes back non-finite (None: raise instead)
    has_semseg_head = property(lambda self: self._model.semseg_head is not None)
    mask_scale = property(lambda self: self._model.semseg_output_scale)                 # inference_model.py:43-45

    @staticmethod
    def load_images(image_paths):
        """BGR uint8 arrays like ``cv2.imread(path, cv2.IMREAD_COLOR)`` (inference_model.py:51-53); PIL when cv2 is absent."""
        try:
            import cv2
            return [cv2.imread(p, cv2.IMREAD_COLOR) for p in image_paths]
        except ImportError:
            from PIL import Image
            return [np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]) for p in image_paths]

    def load_checkpoint_state(self, sd):
        """Reference checkpoints (``torch.load(path)['model']``, inference_model.py:24) carry training-only entries (loss
        buffers) next to the weights: unexpected keys are ignored, but every parameter of this model must be present --
        a key-name drift would otherwise leave random weights behind silently."""
        res = self._model.load_state_dict(sd, strict=False)
        if res.missing_keys:
            raise KeyError("checkpoint lacks %d parameter(s) of the model, e.g. %s" % (len(res.missing_keys), res.missing_keys[:5]))
        return res

    def set_lane(self, lane):
        """Every cached workspace (encoder, decoders, zero-haloed FPN buffers) exists once per lane, so steps enqueued on
        different streams under different lanes do not share scratch memory; the weights are shared."""
        m = self._model
        self.lane = int(lane)
        for mod in (m.backbone, m.embedding_head, m.seediness_head, m.semseg_head):
            if mod is not None:
                mod.lane = self.lane

    def set_precision(self, precision):
        """MFMA mode of every convolution: 'f16x3' (default), 'bf16x6', 'f32' (hip.PRECISIONS)."""
        assert precision in hip.PRECISIONS, precision
        m = self._model
        for mod in (m.backbone, m.embedding_head, m.seediness_head, m.semseg_head):
            if mod is not None:
                mod.precision = precision

    def check_workspaces(self):
        """Debug check (synchronises the device): guard words behind every slice of every cached encoder / decoder workspace that no
        longer hold their canary -- 0 unless a kernel wrote outside its slice.  -> (n_bad, details)"""
        m = self._model
        bad, where = 0, []
        for n_ in ("backbone", "embedding_head", "seediness_head", "semseg_head"):
            mod = getattr(m, n_)
            if mod is not None:
                b, w = mod.check_workspaces()
                bad += b
                where += [(n_,) + x for x in w]
        return bad, where

    def precisions(self):
        """{module name: its MFMA mode} -- to put back with ``restore_precisions`` after a fallback re-run."""
        m = self._model
        return {n_: getattr(m, n_).precision for n_ in ("backbone", "embedding_head", "seediness_head", "semseg_head") if getattr(m, n_) is not None}

    def restore_precisions(self, before):
        for n_, v in before.items():
            getattr(self._model, n_).precision = v

    # ---- one clip ----------------------------------------------------------------------------------
    def _pad_block(self, T, H, W, dev, n, use_last=False):
        """>= n sets ("slots") of four zero-haloed buffers [256][T+2][h+2][pitch] (halos stay zero), one set per clip of an encoder pass.
        The slots of a block are ONE allocation per level, a fixed stride apart: what a clip-batched decoder call needs
        (StemsegDecoderDesc.feat_clip_stride).  A request for more slots than the current block holds makes a larger block current;
        earlier blocks stay alive (captured graphs replay on them).  ``use_last``: a READER of the lane's last encoder pass (the semseg
        helpers) -- it gets the block that pass wrote (``_mark_written``), which is not the current one when a graph captured on a smaller
        block has just been replayed (reading the current block would silently see stale or zero FPN maps)."""
        key = (T, H, W, dev.index, self.lane)
        blk = self._last_written.get(key) if use_last else None      # (the block the lane's last encoder pass wrote: see below)
        if blk is not None and blk["n"] >= n:
            return blk
        blk = self._pads.get(key)
        if blk is None or blk["n"] < n:
            Cn = self._model.backbone.out_channels
            levels = [hip.alloc_padded_batch(n, Cn, T, H // s, W // s, dev) for s in (32, 16, 8, 4)]
            blk = dict(n=n, pads=[[(levels[k][0][c], levels[k][1]) for k in range(4)] for c in range(n)], strides=[levels[k][2] for k in range(4)])
            self._pad_blocks.append(blk)
            self._pads[key] = blk
        return blk

    def _padded_feature_buffers(self, T, H, W, dev, slot=0, use_last=False):
        """The four buffers of ``slot`` in the current block (see _pad_block)."""
        return self._pad_block(T, H, W, dev, slot + 1, use_last)["pads"][slot]

    def _mark_written(self, T, H, W, dev, blk):
        self._last_written[(T, H, W, dev.index, self.lane)] = blk

    @torch.no_grad()
    def embed_clip(self, feature_maps, T, H, W):
        """feature_maps: list over the clip's T slots of dicts {4,8,16,32: [256,h,w]} (device).  Returns the
        clip's (embeddings [E,T,h4,w4], bandwidths [Ev,...] already exp()*10, seediness [1,...])."""
        hip.require_gpu()
        m = self._model
        dev = feature_maps[0][4].device
        blk = self._pad_block(T, H, W, dev, 1)
        pads = blk["pads"][0]
        self._mark_written(T, H, W, dev, blk)
        Cn = m.backbone.out_channels
        for (buf, g), s in zip(pads, (32, 16, 8, 4)):
            stack = torch.stack([fm[s] for fm in feature_maps], 1)                      # [C,T,h,w]  (layout 0)
            hip.copy_to_volume(stack.contiguous(), 0, hip.padded_interior_view(buf, g, Cn, T, H // s, W // s))
        return self._run_heads(pads, T, H, W, dev)

    @torch.no_grad()
    def embed_frames(self, frames):
        """Whole clip at once: frames float32 [T,3,H,W] on the device -> encoder writes its four FPN maps straight into
        the decoders' zero-haloed inputs -> heads.  (No feature cache: used when clips do not share frames.)"""
        hip.require_gpu()
        m = self._model
        T, _, H, W = frames.shape
        dev = frames.device
        blk = self._pad_block(T, H, W, dev, 1)
        pads = blk["pads"][0]
        self._mark_written(T, H, W, dev, blk)
        Cn = m.backbone.out_channels
        vols = {s: hip.padded_interior_view(buf, g, Cn, T, H // s, W // s) for (buf, g), s in zip(pads, (32, 16, 8, 4))}
        m.backbone.run_backbone_into(frames, [vols[s] for s in (4, 8, 16, 32)])
        return self._run_heads(pads, T, H, W, dev)

    @torch.no_grad()
    def embed_frames_batch(self, frames, n_clips):
        """``n_clips`` clips in one encoder pass: frames float32 [n_clips * T, 3, H, W] -> list of (emb, bw, seed) per clip.
        The encoder treats frames independently, so stacking clips only makes its small-map launches (layer3, layer4: 0.4
        waves of the chip per clip) fuller; the decoders and everything after run per clip."""
        hip.require_gpu()
        m = self._model
        NT, _, H, W = frames.shape
        assert NT % n_clips == 0
        T, dev, Cn = NT // n_clips, frames.device, m.backbone.out_channels
        blk = self._pad_block(T, H, W, dev, n_clips)
        self._mark_written(T, H, W, dev, blk)
        pads = blk["pads"]
        vols = []
        for c in range(n_clips):
            v = {s: hip.padded_interior_view(buf, g, Cn, T, H // s, W // s) for (buf, g), s in zip(pads[c], (32, 16, 8, 4))}
            vols += [v[s] for s in (4, 8, 16, 32)]
        m.backbone.run_backbone_into(frames, vols)
        if not self.batch_decoders:
            return [self._run_heads(pads[c], T, H, W, dev, slot=c) for c in range(n_clips)]
        return self._run_heads(pads[0], T, H, W, dev, batch=(n_clips, blk["strides"]))

    @torch.no_grad()
    def embed_frames_windows(self, frames, n_clips, clip_frames, clip_stride):
        """``n_clips`` OVERLAPPING clips in one encoder pass: frames float32 [(n_clips - 1) * clip_stride + clip_frames, 3, H, W]
        (a run of consecutive sequence frames), clip c = frames [c * clip_stride, c * clip_stride + clip_frames) -- the windows
        inference/main.py:23-49 cuts.  A frame shared by two clips goes through the encoder trunk once (what the reference's
        cross-clip feature cache does, inference_model.py:83-108); only the per-clip FPN output convs and everything after
        run per clip.  -> list of (emb, bw, seed)."""
        hip.require_gpu()
        m = self._model
        NT, _, H, W = frames.shape
        assert NT == (n_clips - 1) * clip_stride + clip_frames
        T, dev, Cn = clip_frames, frames.device, m.backbone.out_channels
        blk = self._pad_block(T, H, W, dev, n_clips)
        self._mark_written(T, H, W, dev, blk)
        pads = blk["pads"]
        vols = []
        for c in range(n_clips):
            v = {s: hip.padded_interior_view(buf, g, Cn, T, H // s, W // s) for (buf, g), s in zip(pads[c], (32, 16, 8, 4))}
            vols += [v[s] for s in (4, 8, 16, 32)]
        m.backbone.run_backbone_into(frames, vols, window=(clip_frames, clip_stride))
        if not self.batch_decoders:
            return [self._run_heads(pads[c], T, H, W, dev, slot=c) for c in range(n_clips)]
        return self._run_heads(pads[0], T, H, W, dev, batch=(n_clips, blk["strides"]))

    @torch.no_grad()
    def _run_heads(self, pads, T, H, W, dev, slot=0, batch=None):
        """Both decoders + heads on the clip whose features sit in ``pads`` -> (emb, bw, seed).  ``batch = (n, strides)``: ``pads`` is
        slot 0 of a block (_pad_block) and the n clips of the block go through every decoder stage in ONE launch (the decoders'
        clip batch) -> list of n (emb, bw, seed)."""
        if batch is not None:
            emb, bw, seed = self._run_heads_impl(pads, T, H, W, dev, batch)
            return [(emb[c], bw[c], seed[c]) for c in range(batch[0])]
        return self._run_heads_impl(pads, T, H, W, dev, None)

    def _run_heads_impl(self, pads, T, H, W, dev, batch):
        m = self._model
        feats = ([b for b, _ in pads], (T, H // 4, W // 4))
        r = int(self.resize_scale)

        def resized(seed):
            if batch is None:
                return hip.upsample_trilinear(seed.contiguous(), 1, r, r)
            return torch.stack([hip.upsample_trilinear(seed[c].contiguous(), 1, r, r) for c in range(batch[0])], 0)
        ch = (lambda t, a, b: t[a:b]) if batch is None else (lambda t, a, b: t[:, a:b])
        eh = m.embedding_head
        eh.fuse_bandwidth_activation = True                                             # inference_model.py:148 fused
        seed = None
        eh.concurrency = 1 if self.overlap_decoders else 0
        if eh.seediness_channels == 0 and not self.overlap_decoders:
            m.seediness_head.concurrency, m.seediness_head.detached = 0, False
            seed = m.seediness_head.forward_single(feats, 2, clip_batch=batch)
            if self.resize_scale != 1.0:
                seed = resized(seed)
            main = None
        elif eh.seediness_channels == 0:
            # the seediness decoder shares nothing with the embedding decoder but its (read-only) inputs: enqueue it
            # detached on the library's second stream set, enqueue the embedding decoder, then join -- the two decoders
            # (2 big + 12 small convolutions) fill the chip together
            assert m.seediness_head is not None
            sh = m.seediness_head
            sh.concurrency, sh.detached = 2, True
            seed = sh.forward_single(feats, 2, clip_batch=batch)
            main = sh
        out = eh.forward_single(feats, 2, clip_batch=batch)
        E, Ev = eh.embedding_size, eh.variance_channels
        emb, bw = ch(out, 0, E), ch(out, E, E + Ev)
        if seed is None:
            seed = ch(out, E + Ev, out.shape[0 if batch is None else 1])
        elif main is not None:
            main.join()
            main.detached = False
            if self.resize_scale != 1.0:                                                # inference_model.py:156 quirk
                seed = resized(seed)
        return emb, bw, seed

    @torch.no_grad()
    def semseg_logits_clip(self, T, H, W, dev, slot=0, resize=True):
        """Class logits [C, T, h4*r, w4*r] of the clip whose features sit in the zero-haloed buffers of ``slot`` (inference_model.py:121-124);
        ``resize=False``: at the head's own resolution."""
        sh = self._model.semseg_head
        sh.concurrency, sh.detached = (1 if self.overlap_decoders else 0), False
        logits = sh.forward_single(([b for b, _ in self._padded_feature_buffers(T, H, W, dev, slot=slot, use_last=True)], (T, H // 4, W // 4)), 2)
        if self.resize_scale != 1.0 and resize:
            logits = hip.upsample_trilinear(logits.contiguous(), 1, int(self.resize_scale), int(self.resize_scale))
        return logits.contiguous()

    @torch.no_grad()
    def semseg_logits_batch(self, T, H, W, dev, n_clips, resize=True):
        """``semseg_logits_clip`` for the ``n_clips`` clips of the encoder pass just run (slots 0 .. n_clips - 1 of the current block): the
        third decoder takes them in one launch per stage (the decoders' clip batch; each clip's logits are bit-identical to its own
        call).  -> list of [C, T, h4*r, w4*r]."""
        blk = self._pad_block(T, H, W, dev, n_clips, use_last=True)
        sh = self._model.semseg_head
        sh.concurrency, sh.detached = (1 if self.overlap_decoders else 0), False
        logits = sh.forward_single(([b for b, _ in blk["pads"][0]], (T, H // 4, W // 4)), 2, clip_batch=(n_clips, blk["strides"]))
        r = int(self.resize_scale)
        return [(hip.upsample_trilinear(logits[c].contiguous(), 1, r, r) if (r != 1 and resize) else logits[c]).contiguous() for c in range(n_clips)]

    def semseg_fg_logits_clip(self, T, H, W, dev, slot=0):
        """The channels of the clip's class logits that decide foreground (inference_model.py:207-225), at the head's resolution:
        [1, T, h4, w4] (the last channel of a multi-class head) or [2, T, h4, w4] (a binary head) -- what the clip-parallel
        sequence path exchanges instead of all class logits (pipeline.run_sequence_sharded)."""
        logits = self.semseg_logits_clip(T, H, W, dev, slot=slot, resize=False)
        return logits if logits.shape[0] == 2 else logits[-1:].contiguous()

    @staticmethod
    def _accumulate_semseg(acc, counts, logits, sub):
        """semseg_logits[t][0] += logits[i]; [1] += 1 for every slot i of the clip, duplicates included (:126-128): one
        launch per 'occurrence rank' so that no launch adds twice into the same frame and per-frame order stays slot order."""
        seen, rank = {}, []
        for t in sub:
            rank.append(seen.get(t, 0))
            seen[t] = rank[-1] + 1
            counts[t] += 1
        for r in range(max(rank) + 1):
            hip.semseg_accumulate(acc, logits, [t if k == r else -1 for t, k in zip(sub, rank)])

    @torch.no_grad()
    def get_semseg_masks(self, acc, counts):
        """(fg_masks [F,h,w] float, multiclass_masks) from the accumulated logits (inference_model.py:197-231); with a
        2-channel head the reference crashes on the empty multiclass list (:231) -- an empty list is returned instead."""
        if acc is None:
            return [], []
        cnt = torch.as_tensor(counts, dtype=torch.float32).to(acc.device)
        fg, mc = hip.semseg_masks(acc, cnt, self.semseg_output_type)
        if self.outputs_on_cpu or self.semseg_outputs_on_cpu:
            fg, mc = fg.cpu(), (mc.cpu() if mc is not None else None)
        return fg, (mc if mc is not None else [])

    def _side_stream(self, dev):
        if getattr(self, "_side", None) is None:
            self._side = {}
        if dev.index not in self._side:
            self._side[dev.index] = torch.cuda.Stream(device=dev)
        return self._side[dev.index]

    @torch.no_grad()
    def forward(self, images, subseq_idxes):
        """images: list/array of uint8 BGR frames [H0,W0,3] (or a pre-processed float tensor [N,3,H,W] on the device);
        subseq_idxes: list of frame-index lists (duplicates allowed, inference/main.py:37-39)."""
        hip.require_gpu()
        _config.refresh()
        m = self._model
        dev = next(m.parameters()).device
        if torch.is_tensor(images) and images.dtype == torch.float32 and images.dim() == 4:
            frames = images if images.is_cuda else images.to(dev)
        else:
            if len(images) and isinstance(images[0], (str, bytes)):                     # file paths, as inference/main.py:137-138 passes
                images = self.load_images(images)
            frames, _ = preprocess_frames(np.stack([np.asarray(im) for im in images], 0), dev)
        H, W = frames.shape[-2:]
        cache, deps = {}, {}
        for i, sub in enumerate(subseq_idxes):
            for t in sub:
                deps.setdefault(t, set()).add(i)
        maps, overflow, clip_fg_logits = [], [], []
        acc, counts = None, [0] * len(frames)
        for i, sub in enumerate(subseq_idxes):
            need = sorted(set(t for t in sub if t not in cache))
            if need:
                feats = m.backbone.forward_channel_major(frames[need])                  # one batch for all new frames
                for j, t in enumerate(need):
                    cache[t] = {s: f[:, j] for s, f in zip((4, 8, 16, 32), feats)}
            emb, bw, seed = self.embed_clip([cache[t] for t in sub], len(sub), H, W)
            overflow.append(hip.overflow_status([emb.contiguous(), bw.contiguous(), seed.contiguous()]))
            if self.has_semseg_head:                                                    # same zero-haloed inputs, third decoder
                lo = self.semseg_logits_clip(len(sub), H, W, emb.device, resize=False)
                logits = lo if self.resize_scale == 1.0 else hip.upsample_trilinear(lo, 1, int(self.resize_scale), int(self.resize_scale)).contiguous()
                clip_fg_logits.append(lo if lo.shape[0] == 2 else lo[-1:].contiguous())  # (what the clip-parallel path exchanges)
                overflow.append(hip.overflow_status([logits]))
                if acc is None:
                    acc = torch.zeros((len(frames), logits.shape[0]) + tuple(logits.shape[2:]), dtype=torch.float32, device=logits.device)
                self._accumulate_semseg(acc, counts, logits, sub)
            uniq = sorted(set(sub))
            if len(uniq) != len(sub):                                                   # dict semantics of :137-138: last slot wins
                sel = torch.as_tensor([max(j for j, v in enumerate(sub) if v == t) for t in uniq], device=emb.device)
                emb, bw, seed = emb[:, sel], bw[:, sel], seed[:, sel]
            if self.outputs_on_cpu:
                emb, bw, seed = emb.cpu(), bw.cpu(), seed.cpu()
            maps.append(EmbeddingMapEntry(uniq, emb, bw, seed))
            for t in list(deps):                                                        # evict (:164-173)
                deps[t].discard(i)
                if not deps[t]:
                    cache.pop(t, None)
                    del deps[t]
        # Overflow policy (one 4-byte read per sequence): the split convolution modes have a finite operand range (f16x3: |activation|
        # < 2.6e5) and answer an overflow with non-finite outputs, which every ReLU / pool on the way keeps.  A sequence whose head
        # outputs are not all finite is re-run ONCE in bf16x6 (fp32's exponent range); never hand NaN maps to the clusterer.
        if overflow and bool(torch.cat([o.reshape(-1) for o in overflow]).any().item()):
            before = self.precisions()
            if self.overflow_fallback is None or all(v == self.overflow_fallback for v in before.values()):
                raise hip.NonFiniteError("head outputs hold inf / NaN (convolution mode %s)" % sorted(set(before.values())))
            self.set_precision(self.overflow_fallback)
            try:
                return self.forward(frames, subseq_idxes)
            finally:
                self.restore_precisions(before)
        fg_masks, multiclass_masks = self.get_semseg_masks(acc, counts)
        return {"fg_masks": fg_masks, "multiclass_masks": multiclass_masks, "embeddings": maps, "clip_fg_logits": clip_fg_logits}
