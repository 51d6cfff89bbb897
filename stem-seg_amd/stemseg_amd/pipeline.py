"""Clip-level hot path (embed + cluster) and its multi-GPU sharding.

* ``ClipPipeline.step`` is BASELINE.json's unit of work: one T-frame clip through encoder -> two 3-D decoders ->
  fused heads -> fg mask -> fg gather -> sequential clustering, all enqueued on one HIP stream with no host
  synchronisation (N, K, the instance list and the overflow flags stay on the device until the caller reads them).
* ``run_sequence_sharded`` is the one-process-per-GPU form for long sequences, partitioned as SURVEY.md section 8(e) lays out:
  every rank takes a contiguous block of the clips, embeds it (frames shared by neighbouring clips of the block pass the encoder
  trunk once) AND clusters it with label_start = 1; two small all-gathers (RCCL over xGMI on the GPU box, gloo in the CPU tests)
  carry the foreground evidence (seediness planes, or the semseg head's foreground probability) and then one byte per voxel of
  clip-local label codes + the clustering records; the Hungarian chain runs on label-pair tables, so every rank ends with the
  single-process result bit for bit -- embeddings included: the encoder plans its launches for a fixed frame count
  (ResNetFPN.plan_frames), so a clip's maps do not depend on how many clips shared its encoder pass.
"""
import torch

from . import hip
from .config import cfg
from .inference.main import TrackGenerator, fg_masks_from_seediness, get_subsequence_frames  # noqa: F401
from .modeling.inference_model import EmbeddingMapEntry, InferenceModel


class ClipPipeline(object):
    def __init__(self, model=None, seediness_thresh=0.25, device="cuda"):
        self.model = model if model is not None else InferenceModel()
        self.model.to(device)
        self.tg = TrackGenerator(self.model, "davis", seediness_thresh=seediness_thresh)
        self.clusterer = self.tg.chainer.clusterer
        self.seediness_thresh = seediness_thresh
        self.batch_clustering = True         # step_batch: the clips of a step share the clustering launches (False: one sequence per clip)

    @torch.no_grad()
    def embed(self, frames):
        """frames: float32 [T,3,H,W] on the device (pre-processed, H and W multiples of 32)."""
        return self.model.embed_frames(frames.contiguous())

    @torch.no_grad()
    def gather(self, emb, bw, seed, fg=None):
        """Single-clip fg mask (seediness > thr unless ``fg`` is given) + gather; device tensors only."""
        if fg is None:
            acc = torch.empty_like(seed[0])
            hip.seediness_accumulate(acc, seed[0].contiguous(), True)
            fg = hip.fg_mask(acc, 1.0, self.seediness_thresh)
        e, b, s, vox, offs = hip.fg_gather(emb.contiguous(), bw.contiguous(), seed.contiguous(), fg)
        return dict(points=(e, b, s, offs[seed.shape[1]:]), voxel_index=vox, frame_offsets=offs, fg=fg)

    @torch.no_grad()
    def cluster(self, emb, bw, seed, label_start=1, fg=None):
        """Single-clip fg mask (seediness > thr unless ``fg`` is given), gather, clustering.  Returns device tensors only."""
        g = self.gather(emb, bw, seed, fg)
        e, b, s, n = g.pop("points")
        labels, meta_dev, _, _ = self.clusterer.enqueue(e, b, s, label_start, n)
        g.update(labels=labels, meta=meta_dev)
        return g

    def _finish_clip(self, emb, bw, seed, T, H, W, slot=0, defer_clustering=False, logits=None):
        """Everything after the embedding decoder of one independent clip.  Presets with a semseg head (YouTube-VIS, KITTI-MOTS):
        third decoder -> class logits (x resize_scale, inference_model.py:121-124) -> foreground = its fg probability > 0.5
        (inference_model.py:197-231, inference/main.py:142-144); under --resize_embeddings the head outputs are up-sampled x4 and
        the clip is clustered at full resolution (online_chainer.py:127-140)."""
        fg = None
        emb0, bw0, seed0 = emb, bw, seed                     # the decoders' own outputs (before any resize)
        if self.model.has_semseg_head:
            if logits is None:                               # (step_batch hands over the clip's share of a batched third-decoder call)
                logits = self.model.semseg_logits_clip(T, H, W, emb.device, slot=slot)
            fg, _ = hip.semseg_fg_clip(logits, 0.5)
        r = int(self.model.resize_scale)
        if r != 1:
            emb, bw = hip.upsample_trilinear(emb.contiguous(), 1, r, r), hip.upsample_trilinear(bw.contiguous(), 1, r, r)
            if seed.shape[-1] != emb.shape[-1]:          # (a separate seediness head is already resized by the model, :156)
                seed = hip.upsample_trilinear(seed.contiguous(), 1, r, r)
        # overflow guard: a non-finite head output (an operand left the split convolution mode's range) is flagged on the device and
        # read with the clustering record -- such maps are never clustered silently (hip.read_cluster_meta raises NonFiniteError)
        status = hip.overflow_status([t.contiguous() for t in ([emb0, bw0, seed0] + ([logits] if logits is not None else []))])
        out = self.gather(emb, bw, seed, fg=fg) if defer_clustering else self.cluster(emb, bw, seed, fg=fg)
        out.update(emb=emb, bw=bw, seed=seed, status=status)
        if logits is not None:
            out["semseg_logits"] = logits
        return out

    @torch.no_grad()
    def step(self, frames):
        T, _, H, W = frames.shape
        emb, bw, seed = self.embed(frames)
        return self._finish_clip(emb, bw, seed, T, H, W)

    @torch.no_grad()
    def step_checked(self, frames, fallback_precision="bf16x6"):
        """``step`` + the consumer's read-back, with the overflow policy: when a head output of the clip is non-finite (an activation
        left the range of the split convolution mode, e.g. |a| >= 2.6e5 in f16x3) the clip is re-run ONCE in ``fallback_precision``
        (bf16x6: fp32's full exponent range, twice the matrix work) and the model's mode is restored; still non-finite -> raises
        hip.NonFiniteError.  -> (step dict, StemsegClusterMeta on the host)."""
        out = self.step(frames)
        try:
            return out, hip.read_cluster_meta(out["meta"], out["status"])
        except hip.NonFiniteError:
            before = self.model.precisions()
            if all(v == fallback_precision for v in before.values()):
                raise
            self.model.set_precision(fallback_precision)        # (both packings stay cached: modules key them by precision)
            try:
                out = self.step(frames)
                return out, hip.read_cluster_meta(out["meta"], out["status"])
            finally:
                self.model.restore_precisions(before)

    @torch.no_grad()
    def step_batch(self, frames, n_clips):
        """``n_clips`` clips stacked along the frame axis ([n_clips * T, 3, H, W]): one encoder pass, then decoders + fg mask +
        gather + clustering per clip.  Returns a list of ``step``-style dicts."""
        NT, _, H, W = frames.shape
        # (with a semseg head its decoder reads clip c's zero-haloed FPN buffers, slot c, which stay valid after the pass)
        heads = self.model.embed_frames_batch(frames.contiguous(), n_clips)
        sem = [None] * n_clips
        if self.model.has_semseg_head and self.model.batch_decoders and n_clips > 1:
            sem = self.model.semseg_logits_batch(NT // n_clips, H, W, frames.device, n_clips)      # the third decoder: all clips per launch
        outs = [self._finish_clip(emb, bw, seed, NT // n_clips, H, W, slot=c, defer_clustering=self.batch_clustering, logits=sem[c])
                for c, (emb, bw, seed) in enumerate(heads)]
        if self.batch_clustering:
            # the clips are independent point sets: ONE sequence of clustering launches serves all of them (grid.y = clip)
            res = self.clusterer.enqueue_batch([o.pop("points") for o in outs], 1)
            for o, (labels, meta_dev) in zip(outs, res):
                o.update(labels=labels, meta=meta_dev)
        return outs

    @torch.no_grad()
    def capture_embed(self, example_frames, n_clips=None, lane=0):
        """hipGraph of the embedding alone (encoder + decoders + heads): for drivers that cluster later, e.g. the sharded
        sequence path, where clustering waits for the cross-clip foreground mask.  ``n_clips``: that many clips stacked along the
        frame axis share the encoder pass (``run`` then returns a list).  The outputs are static (emb, bw, seed) tensors that
        the next replay overwrites."""
        return GraphedStep(self, example_frames, False, n_clips, lane, embed_only=True)

    @torch.no_grad()
    def embed_many(self, frames, clips, batch=4, lanes=2, use_graph=True, share_overlap=True, with_fg_logits=False):
        """Embeds the clips ``clips`` (lists of frame indices into ``frames`` [F,3,H,W], all of one length) ``batch`` at a time
        through one encoder pass each, full batches as hipGraph replays alternating over ``lanes`` streams, the remainder
        eagerly.  When the clips are consecutive windows of the sequence (constant stride, as get_subsequence_frames cuts them)
        and ``share_overlap``, a pass covers the windows' UNION of frames: shared frames go through the encoder trunk once.
        Returns per clip a [E+Ev+1, T, h4, w4] block (emb | bw | seed stacked) that the caller owns; ``with_fg_logits`` (presets with
        a semseg head): 1 or 2 more trailing channels, the foreground logits of the clip at the heads' resolution
        (InferenceModel.semseg_fg_logits_clip) -- the foreground evidence ``run_sequence_sharded`` exchanges."""
        dev = frames.device
        out = [None] * len(clips)
        T = len(clips[0])
        stride = clips[1][0] - clips[0][0] if len(clips) > 1 else T
        # windows = the longest on-stride prefix of the clips (get_subsequence_frames appends an off-stride tail clip whenever
        # (F - T) % (T - overlap) != 0: it is embedded on its own, the prefix still shares the trunk)
        n_win = 0
        if bool(share_overlap) and 0 < stride < T:
            while n_win < len(clips) and list(clips[n_win]) == list(range(clips[0][0] + n_win * stride, clips[0][0] + n_win * stride + T)):
                n_win += 1
        if n_win < 2:
            n_win = 0
        if 0 < n_win < len(clips):
            head = self.embed_many(frames, clips[:n_win], batch, lanes, use_graph, share_overlap, with_fg_logits)
            tail = self.embed_many(frames, clips[n_win:], batch, lanes, use_graph, False, with_fg_logits)
            return head + tail
        windows = n_win == len(clips) and n_win > 0
        groups = [list(range(i, min(i + batch, len(clips)))) for i in range(0, len(clips), batch)]

        def pass_frames(g):                                 # frame indices of one encoder pass
            if windows:
                return list(range(clips[g[0]][0], clips[g[-1]][-1] + 1))
            return sum([list(clips[c]) for c in g], [])

        def embed_pass(x, n):
            if windows and n > 1:
                res = self.model.embed_frames_windows(x, n, T, stride)
            else:
                res = self.model.embed_frames_batch(x, n) if n > 1 else [self.model.embed_frames(x)]
            if with_fg_logits:                               # third decoder on clip c's zero-haloed FPN buffers (slot c)
                H_, W_ = x.shape[-2:]
                res = [r_ + (self.model.semseg_fg_logits_clip(T, H_, W_, x.device, slot=c),) for c, r_ in enumerate(res)]
            return res
        full = [g for g in groups if len(g) == batch] if use_graph and batch > 1 else []
        if full:
            key = (batch, T, stride if windows else 0, tuple(frames.shape[1:]), lanes, bool(with_fg_logits), int(self.model._model.backbone.plan_frames),
                   tuple(sorted(self.model.precisions().items())))
            cache = self.__dict__.setdefault("_embed_graphs", {})
            if key not in cache:
                ex = frames[torch.as_tensor(pass_frames(full[0]), device=dev)].contiguous()
                cache[key] = [GraphedStep(self, ex, False, batch, 10 + k, embed_fn=lambda x: embed_pass(x.contiguous(), batch))
                              for k in range(max(1, lanes))]
            gs = cache[key]
            pending = [None] * len(gs)

            def collect(k):
                if pending[k] is not None:
                    gs[k].wait()
                    for c, parts in zip(pending[k][0], gs[k].out):
                        out[c] = torch.cat(list(parts), 0)
                    pending[k] = None
            for n, g in enumerate(full):
                k = n % len(gs)
                collect(k)                                   # the lane's previous outputs, before the replay overwrites them
                x = frames[torch.as_tensor(pass_frames(g), device=dev)]
                x.record_stream(gs[k].stream)                # the lane's stream copies it into the graph's input: the caching
                gs[k].run_async(x)                           # allocator must not hand the block out again before that copy ran
                pending[k] = (g, x)                          # (and the tensor itself stays referenced until collect(k))
            for k in range(len(gs)):
                collect(k)
        for g in groups:
            if g in full:
                continue
            res = embed_pass(frames[torch.as_tensor(pass_frames(g), device=dev)].contiguous(), len(g))
            for c, parts in zip(g, res):
                out[c] = torch.cat(list(parts), 0)
        return out

    def capture(self, example_frames, overlap=False, n_clips=None, lane=0):
        """Capture ``step`` for clips of ``example_frames``' shape into ONE hipGraph (~330 kernel nodes on a single stream:
        measured, the decoders' fork/join branch streams buy nothing once every conv fills the chip, and single-stream
        capture is the robust form).  Returns a ``GraphedStep``; its outputs are static device tensors overwritten by
        every ``run``.  Requires that ``step`` has no host synchronisation -- which is how the path is built.
        ``lane``: graphs captured under different lanes own disjoint workspaces and replay on their own streams, so several
        steps can be in flight: the kernels of one fill the tail rounds and memory-bound phases of the other (measured
        +6 % with two lanes at 4 clips per step, +9 % at 1)."""
        return GraphedStep(self, example_frames, overlap, n_clips, lane)


class GraphedStep(object):
    def __init__(self, pipe, example_frames, overlap=False, n_clips=None, lane=0, embed_only=False, embed_fn=None):
        self.pipe = pipe
        self.lane = lane
        self.stream = torch.cuda.Stream(device=example_frames.device)       # replays of this lane are ordered on this stream
        pipe.model.set_lane(lane)
        fn = pipe.step if n_clips is None else (lambda x: pipe.step_batch(x, n_clips))     # n_clips: ``run`` returns a list
        if embed_only:
            fn = pipe.embed if n_clips is None else (lambda x: pipe.model.embed_frames_batch(x.contiguous(), n_clips))
        if embed_fn is not None:
            fn = embed_fn
        self._fn, self._overlap = fn, bool(overlap)
        prev = pipe.model.overlap_decoders
        pipe.model.overlap_decoders = bool(overlap)       # True = capture the fork/join branch streams too (experimental)
        try:
            self.static_in = example_frames.clone()
            dev = example_frames.device
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                fn(self.static_in)                        # allocates every cached workspace outside the capture
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: calls made meanwhile by OTHER threads (e.g. the RCCL watchdog's event queries when a process
            # group is alive) must not invalidate this thread's capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = fn(self.static_in)
            torch.cuda.synchronize(dev)
            # the zero-haloed FPN blocks this graph's encoder pass writes (this lane's entries): a replay re-marks them as the lane's last
            # written ones, so an eager semseg call that follows reads what the replay produced (InferenceModel._pad_block, use_last)
            self._written = {k: v for k, v in pipe.model._last_written.items() if k[-1] == lane}
        finally:
            pipe.model.overlap_decoders = prev
            pipe.model.set_lane(0)

    def run(self, frames):
        """Device-to-device copy of the clip into the graph's input, one graph launch; returns the static output dict."""
        self.static_in.copy_(frames, non_blocking=True)
        self.graph.replay()
        self.pipe.model._last_written.update(self._written)
        return self.out

    def run_async(self, frames):
        """Same on this lane's own stream (after whatever the current stream has enqueued so far, e.g. the producer of
        ``frames``); follow with ``wait()`` -- or read the outputs under ``torch.cuda.stream(self.stream)`` -- before use."""
        self.stream.wait_stream(torch.cuda.current_stream(frames.device))
        with torch.cuda.stream(self.stream):
            self.run(frames)
        return self.out

    def wait(self):
        torch.cuda.current_stream(self.static_in.device).wait_stream(self.stream)

    def collect(self, fallback_precision="bf16x6"):
        """The consumer's read-back of this lane's last replay of a clustering step, with the overflow policy of
        ``ClipPipeline.step_checked``: -> (list of step dicts, list of StemsegClusterMeta).  When a head output of one of the clips is
        non-finite (an activation left the split convolution mode's range) the lane's batch -- its frames are still in the graph's
        static input -- is run ONCE more, eagerly, in ``fallback_precision`` on this lane's own stream and workspaces, and those
        results are returned (fresh tensors, not the graph's static ones, allocated on ``self.stream``: consume them under that stream or
        after ``wait()``); the model's mode and lane are restored.  A production lane therefore never hands out NaN maps and never just
        raises (``fallback_precision=None`` restores raising)."""
        outs = self.out if isinstance(self.out, (list, tuple)) else [self.out]
        with torch.cuda.stream(self.stream):
            try:
                return list(outs), [hip.read_cluster_meta(o["meta"], o.get("status")) for o in outs]
            except hip.NonFiniteError:
                model = self.pipe.model
                before = model.precisions()
                if fallback_precision is None or all(v == fallback_precision for v in before.values()):
                    raise
                prev, prev_lane = model.overlap_decoders, model.lane
                model.set_precision(fallback_precision)
                model.set_lane(self.lane)
                model.overlap_decoders = self._overlap
                try:
                    redo = self._fn(self.static_in)
                    redo = list(redo) if isinstance(redo, (list, tuple)) else [redo]
                    return redo, [hip.read_cluster_meta(o["meta"], o.get("status")) for o in redo]
                finally:
                    model.restore_precisions(before)
                    model.overlap_decoders = prev
                    model.set_lane(prev_lane)         # (the caller's lane, not lane 0: its workspaces may be in use by in-flight eager work)


# ------------------------------------------------------------------------------------------------ multi-GPU
def shard_clips(n_clips, rank, world_size):
    """Balanced CONTIGUOUS blocks (sizes differ by at most one): a rank's clips are consecutive windows of the sequence, so the
    frames two of them share go through that rank's encoder trunk once (ClipPipeline.embed_many)."""
    q, rem = divmod(n_clips, world_size)
    start = rank * q + min(rank, rem)
    return list(range(start, start + q + (1 if rank < rem else 0)))


def clip_owner(ci, n_clips, world_size):
    """-> (rank, slot within the rank's block) of clip ``ci`` under shard_clips."""
    q, rem = divmod(n_clips, world_size)
    edge = rem * (q + 1)
    if ci < edge:
        return ci // (q + 1), ci % (q + 1)
    return rem + (ci - edge) // max(q, 1), (ci - edge) % max(q, 1)


@torch.no_grad()
def run_sequence_sharded(n_frames, embed_clip_fn, chainer, dataset_name="davis", frame_overlap=-1, seediness_thresh=0.25,
                         fg_mask_fn=None, group=None, stats=None, embed_many_fn=None, channel_split=None, outputs_on_cpu=True, comm=None,
                         fg_logit_channels=0):
    """One long sequence over the ranks of ``group``, partitioned as SURVEY.md 8(e) lays out:

      1. every rank embeds ITS contiguous block of clips (``embed_clip_fn(frame_indices) -> (emb [E,T,h,w], bw [Ev,T,h,w],
         seed [1,T,h,w])`` per clip, or ``embed_many_fn(list of clips) -> list of stacked [E+Ev+1,T,h,w] blocks`` with
         ``channel_split = (E, Ev)``, e.g. ClipPipeline.embed_many: several overlapping windows per encoder pass);
      2. all-gather #1: the FOREGROUND EVIDENCE only -- the seediness planes (1 of the E+Ev+1 channels) -> the cross-clip
         mean-seediness foreground mask of the whole sequence on every rank (inference/main.py:93-103), one launch; or, for presets
         with a semseg head (``fg_logit_channels`` = 1: the foreground logit of a multi-class head, 2: both logits of a binary
         head; the clip functions then return a fourth tensor / extra trailing channels [Cfg, T, h, w], BEFORE any resize), those
         planes -> every rank resizes them (x ``chainer.resize_scale``), accumulates them per frame in clip order, averages, and
         thresholds the foreground probability at 0.5 (inference_model.py:121-128, 197-231, inference/main.py:142-144);
      3. every rank gathers + clusters its OWN clips with label_start = 1 (labels are i + label_start, clusterers.py:121, so the
         global id is an offset applied later) -- at FULL resolution when ``chainer.resize_scale`` > 1 (embeddings, bandwidths and
         seediness resized first, online_chainer.py:127-140) -- and leaves one byte per voxel (0 background, 1..K instance, 255
         outlier);
      4. all-gather #2: those byte planes + the 4.4 KB clustering record per clip;
      5. replicated and tiny: ONE launch builds the K1 x K2 label-pair tables of every (clip, overlap frame), one read-back, the
         Hungarian chain with ``next_track_label = highest id + 1`` runs on the host over the tables
         (online_chainer.stitch_from_tables = online_chainer.py:193-236, :291-343, :43-49), one launch turns codes into final labels.

    Returns OnlineChainer.process(...)'s structure, identical on every rank and bit-identical to the single-process result
    (tests/test_distributed_cpu.py: world 1 / 2 / 3 / 8 against the reference-generated goldens).  ``outputs_on_cpu=False`` leaves
    the label tensors on the device (the reference's structure holds host tensors).  ``stats`` (dict, optional) receives the
    exchanges' sizes and durations."""
    import time
    import numpy as np
    from .inference.online_chainer import stitch_from_tables
    dist = comm if comm is not None else TorchComm(group)           # (comm: tests drive N "virtual ranks" in one process)
    # the two collectives run whenever there is a group to run them on -- also a group of ONE (a world-1 RCCL group exercises the
    # same device-tensor all-gathers as world 8: tests/test_gpu_nccl.py); without a process group the buffers are used in place
    distributed = bool(getattr(dist, "active", dist.world > 1))
    rank, world = dist.rank, dist.world
    ops, clusterer = chainer.ops, chainer.clusterer
    r_scale = float(chainer.resize_scale)
    Cfg = int(fg_logit_channels)
    clips, _ = get_subsequence_frames(n_frames, cfg.INPUT.NUM_FRAMES, dataset_name, frame_overlap)
    n_clips = len(clips)
    mine = shard_clips(n_clips, rank, world)
    per_rank = (n_clips + world - 1) // world
    T = len(clips[0])
    uniq = [sorted(set(c)) for c in clips]
    sels = [None if len(u) == len(c) else [max(j for j, v in enumerate(c) if v == t) for t in u] for u, c in zip(uniq, clips)]

    # ---- 1. embed this rank's clips ---------------------------------------------------------------------------------
    blocks = []                                           # per own clip: (emb, bw, seed[, fg logits]) with all T slots
    if embed_many_fn is not None and mine:
        E, Ev = channel_split
        blocks = [(b[:E], b[E:E + Ev], b[E + Ev:E + Ev + 1]) + ((b[E + Ev + 1:E + Ev + 1 + Cfg],) if Cfg else ()) for b in embed_many_fn([clips[ci] for ci in mine])]
    elif mine:
        blocks = [tuple(embed_clip_fn(clips[ci])) for ci in mine]
    assert all(len(b) == (4 if Cfg else 3) for b in blocks), "clip functions must return (emb, bw, seed%s)" % (", fg logits" if Cfg else "")

    # ---- 2. all-gather #1: the foreground evidence (seediness planes, or the semseg head's foreground logits) -------------
    shape_info = None
    if blocks:
        seed0 = blocks[0][2]
        dev, (hl, wl) = seed0.device, tuple(blocks[0][0].shape[-2:])
        ev0 = blocks[0][3] if Cfg else seed0               # the planes all-gather #1 carries
        shape_info = [hl, wl, blocks[0][0].shape[0], int(ev0.shape[-2]), int(ev0.shape[-1]), int(seed0.shape[-2]), int(seed0.shape[-1])]
    if distributed and n_clips < world:                   # some ranks own no clip: they learn the map size from the others
        dev = _default_device() if not blocks else dev
        info = torch.tensor(shape_info if shape_info else [0] * 7, dtype=torch.int64, device=dev)
        infos = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(infos, info)
        shape_info = next(i for i in infos if int(i[0]) > 0).tolist()
    hl, wl = int(shape_info[0]), int(shape_info[1])       # the heads' resolution
    E_dims = int(shape_info[2])
    he, we = int(shape_info[3]), int(shape_info[4])       # resolution of the exchanged evidence planes
    # A model with a SEPARATE seediness head resizes its seediness itself (inference_model.py:156; InferenceModel._run_heads), and the
    # chainer then resizes embeddings, bandwidths AND seediness (online_chainer.py:127-140): under --resize_embeddings such a model's
    # seediness ends up at r^2 times the embeddings' resolution and the reference's clustering fails on the size mismatch -- the
    # reference only ever combines --resize_embeddings with in-head seediness (youtube_vis.yaml).  Same here, said clearly, instead of
    # a reshape error or a silently different result.
    seed_hw = (int(shape_info[5]), int(shape_info[6]))
    if seed_hw != (hl, wl):
        raise ValueError("run_sequence_sharded: the seediness planes arrive at %s but the embeddings at %s: a separate seediness head resizes its "
                         "output itself (inference_model.py:156) and the chainer would resize it again (online_chainer.py:127-140) -- the reference "
                         "fails on this combination too; use resize_scale 1 or a model with in-head seediness" % (seed_hw, (hl, wl)))
    Cg = Cfg if Cfg else 1
    seeds_local = torch.zeros((per_rank, Cg, T, he, we), dtype=torch.float32, device=dev)
    for slot, blk in enumerate(blocks):
        seeds_local[slot] = (blk[3] if Cfg else blk[2]).reshape(Cg, T, he, we)
    if distributed:
        seeds_all = [torch.empty_like(seeds_local) for _ in range(world)]
        t_ag1 = _Timer(seeds_local)
        dist.all_gather(seeds_all, seeds_local)        # data-path collective #1 (RCCL over xGMI on the box)
        t_ag1.stop()
    else:
        seeds_all, t_ag1 = [seeds_local], None

    def plane_of(ci):
        return clip_owner(ci, n_clips, world)
    if Cfg:
        # semseg foreground: every slot of every clip (repeats included) adds its resized logits to its frame, clip order
        per_clip = []
        for ci in range(n_clips):
            owner, slot = plane_of(ci)
            per_clip.append((list(clips[ci]), seeds_all[owner][slot]))
        fg = ops.to_device(fg_mask_fn(per_clip, n_frames, r_scale) if fg_mask_fn is not None else ops.fg_from_semseg(per_clip, n_frames, r_scale))
    else:
        entries = []
        for ci in range(n_clips):
            owner, slot = plane_of(ci)
            sd = seeds_all[owner][slot][0]
            if sels[ci] is not None:
                sd = sd[torch.as_tensor(sels[ci], device=sd.device)]
            if r_scale != 1.0 and fg_mask_fn is None:      # (a seediness-derived mask is formed at the resolution the clusterer sees)
                sd = ops.resize(sd[None], r_scale)[0]
            entries.append(EmbeddingMapEntry(uniq[ci], None, None, sd[None]))
        fg_fn = fg_mask_fn if fg_mask_fn is not None else fg_masks_from_seediness
        fg = ops.to_device(fg_fn(entries, seediness_thresh))              # [F, h, w] uint8
    h, w = int(fg.shape[-2]), int(fg.shape[-1])
    assert (h, w) == (int(round(hl * r_scale)), int(round(wl * r_scale))), \
        "Size mismatch between embeddings {} (x {}) and masks {}".format((hl, wl), r_scale, tuple(fg.shape))
    hw = h * w
    vox_all, offs_all = ops.compact(fg)

    # ---- 3. cluster this rank's clips with label_start = 1; one byte per voxel ------------------------------------------
    meta_planes = (ops.meta_bytes() + 1 + hw - 1) // hw                # clustering record + one overflow byte per clip
    P = T + meta_planes                                               # planes per clip in the exchange buffer
    codes_local = torch.zeros((per_rank, P * hw), dtype=torch.uint8, device=dev)
    for slot, ci in enumerate(mine):
        emb, bw, seed = blocks[slot][:3]
        if sels[ci] is not None:
            sel = torch.as_tensor(sels[ci], device=emb.device)
            emb, bw, seed = emb[:, sel], bw[:, sel], seed[:, sel]
        if r_scale != 1.0:                                   # online_chainer.py:127-140: x r trilinear of all three
            emb, bw, seed = ops.resize(emb, r_scale), ops.resize(bw, r_scale), ops.resize(seed, r_scale)
        fg_clip = fg[torch.as_tensor(uniq[ci], device=fg.device)]
        pts = ops.gather(emb, bw, seed, fg_clip)
        labels, meta_dev, _ = ops.cluster(clusterer, pts, 1, False)
        ops.codes_from_labels(pts, labels, 1, codes_local[slot, :len(uniq[ci]) * hw])
        codes_local[slot, T * hw:T * hw + ops.meta_bytes()] = ops.pack_meta(meta_dev)
        # overflow guard: one byte per clip travels with the record, so EVERY rank refuses a sequence with a non-finite head output
        codes_local[slot, T * hw + ops.meta_bytes()] = ops.overflow_byte(blocks[slot])

    # ---- 4. all-gather #2: label codes + clustering records -------------------------------------------------------------
    if distributed:
        codes_all = torch.empty((world, per_rank, P * hw), dtype=torch.uint8, device=dev)
        t_ag2 = _Timer(codes_local)
        dist.all_gather_into_tensor(codes_all, codes_local)            # data-path collective #2
        t_ag2.stop()
    else:
        codes_all, t_ag2 = codes_local[None], None
    planes = codes_all.view(world * per_rank * P, hw)

    def plane_index(ci, j):
        owner, slot = plane_of(ci)
        return (owner * per_rank + slot) * P + j

    # ---- 5. tables for every (clip, frame), one read-back, host chain, final labels -------------------------------------
    from .inference.online_chainer import frame_sources
    src, _, _ = frame_sources(uniq)
    item_of, plane_a, plane_b = {}, [], []
    for ci, frames in enumerate(uniq):
        for j, t in enumerate(frames):
            item_of[(ci, j)] = len(plane_b)
            c, js = src[t]
            plane_a.append(-1 if (c, js) == (ci, j) else plane_index(c, js))
            plane_b.append(plane_index(ci, j))
    B = clusterer.max_instances + 2
    tables_dev = ops.pair_tables(planes, plane_a, plane_b, B)
    meta_rows = codes_all.view(world * per_rank, P * hw)[:, T * hw:T * hw + ops.meta_bytes() + 1]
    tables, offs, meta_raw = ops.read_back(tables_dev, offs_all, meta_rows.contiguous())
    metas = []
    for ci in range(n_clips):
        owner, slot = plane_of(ci)
        row = meta_raw[owner * per_rank + slot]
        if int(row[-1]) != 0:
            raise hip.NonFiniteError("clip %d (rank %d): a head output holds inf / NaN (an operand left the convolution mode's range) -- "
                                     "re-run the sequence with precision 'bf16x6'" % (ci, owner))
        m_ = ops.unpack_meta(row[:-1].tobytes())
        # (labels are i + label_start with i < K: with K <= max_instances every code is <= B - 2, so the out-of-range clamp of
        # labels_to_codes_kernel cannot be reached -- checked here rather than trusted)
        assert int(m_.K) <= clusterer.max_instances, "clip %d: K = %d exceeds max_instances = %d" % (ci, int(m_.K), clusterer.max_instances)
        metas.append(m_)
    t_host = time.perf_counter()
    st = stitch_from_tables(uniq, tables, item_of, [m.K for m in metas], B)
    host_ms = 1e3 * (time.perf_counter() - t_host)

    offs = [int(v) for v in offs]
    n_total = offs[-1]
    items, luts = [], []
    for t in range(n_frames):                                           # the track container's frames
        assert t in st["src"], "frame %d is in no clip" % t
        c, js = st["src"][t]
        items.append((offs[t], offs[t + 1] - offs[t], t * hw, plane_index(c, js), offs[t]))
        luts.append(st["lut_track"][(c, js)])
    cursor = n_total
    sub_slices = []
    for ci, frames in enumerate(uniq):                                  # per-clip label lists (relabelled on the frames the clip adds)
        row = []
        for j, t in enumerate(frames):
            n_t = offs[t + 1] - offs[t]
            items.append((offs[t], n_t, t * hw, plane_index(ci, j), cursor))
            luts.append(st["lut_sub"][(ci, j)])
            row.append((cursor, cursor + n_t))
            cursor += n_t
        sub_slices.append(row)
    max_count = max([it[1] for it in items] + [0])
    out = ops.labels_from_codes(planes, vox_all, np.asarray(items, np.int64).reshape(-1, 5), np.stack(luts), max_count, cursor)
    local = vox_all[:n_total].long() - torch.repeat_interleave(
        torch.arange(n_frames, device=vox_all.device) * hw, torch.as_tensor([offs[t + 1] - offs[t] for t in range(n_frames)], device=vox_all.device))
    ys, xs = torch.div(local, w, rounding_mode="floor"), local % w
    conv = (lambda x: x.cpu()) if outputs_on_cpu else (lambda x: x)
    out_h, ys, xs = conv(out), conv(ys), conv(xs)
    track = [out_h[offs[t]:offs[t + 1]] for t in range(n_frames)]
    mask_idxes = [(ys[offs[t]:offs[t + 1]], xs[offs[t]:offs[t + 1]]) for t in range(n_frames)]
    subseq_labels = [[out_h[a:b] for a, b in row] for row in sub_slices]
    # per-id point counts / lifetimes (TrackContainer.get_track_mask_idxes, online_chainer.py:94-117) from the tables
    from collections import defaultdict
    counts, first, last = defaultdict(lambda: 0), {}, {}
    for t in range(n_frames):
        c, js = st["src"][t]
        per_bin = tables[item_of[(c, js)]].sum(0)
        ids = st["lut_track"][(c, js)]
        agg = {}
        for b in np.flatnonzero(per_bin[1:]) + 1:
            agg[int(ids[b])] = agg.get(int(ids[b]), 0) + int(per_bin[b])
        for k, n in agg.items():
            counts[k] += n
            first[k] = min(first.get(k, 10000), t)
            last[k] = max(last.get(k, -1), t)
    lifetimes = {k: last[k] - first[k] for k in first}
    subseq_meta = []
    for ci in range(n_clips):
        info = clusterer.meta_to_dict(metas[ci], E_dims, st["label_start"][ci])
        info["instance_labels"] = st["instance_labels"][ci]
        subseq_meta.append(info)
    if stats is not None:
        ag1_ms, ag2_ms = (t_ag1.elapsed() if t_ag1 else 0.0), (t_ag2.elapsed() if t_ag2 else 0.0)     # (events long complete: no extra sync)
        stats.update(allgather_ms=ag1_ms + ag2_ms, allgather_seediness_ms=ag1_ms, allgather_codes_ms=ag2_ms, host_chain_ms=host_ms,
                     allgather_bytes=(seeds_local.numel() * 4 + codes_local.numel()) * (world - 1) if world > 1 else 0,
                     n_clips=n_clips, clips_this_rank=len(mine), world=world, collectives_run=2 if distributed else 0,
                     backend=getattr(dist, "backend", None), comm_calls=dict(getattr(dist, "calls", {})))
    return (track, counts, lifetimes), mask_idxes, subseq_labels, [], subseq_meta


class TorchComm(object):
    """The two collectives of the sharded path on torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, gloo in the
    CPU tests); world 1 when no process group is initialised."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        on = dist.is_available() and dist.is_initialized()
        self.active = on                                              # a process group exists (possibly of one rank)
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.backend = dist.get_backend(group) if on else None
        self.calls = {"all_gather": 0, "all_gather_into_tensor_nccl": 0, "all_gather_into_tensor_list": 0}

    def all_gather(self, outs, t):
        self.calls["all_gather"] += 1
        self.dist.all_gather(outs, t.contiguous(), group=self.group)

    def all_gather_into_tensor(self, out, t):
        if t.is_cuda and self.backend == "nccl":
            self.calls["all_gather_into_tensor_nccl"] += 1
            self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        else:                                                          # (gloo: list form)
            self.calls["all_gather_into_tensor_list"] += 1
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t.contiguous(), group=self.group)
            for r, p_ in enumerate(parts):
                out[r] = p_


class _Timer(object):
    """Elapsed time of a collective: device events on the GPU (recorded on the caller's stream, read later -- no host
    synchronisation of its own), wall clock on the host."""

    def __init__(self, t):
        import time
        self.gpu = t.is_cuda
        if self.gpu:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            self.t0 = time.perf_counter()

    def stop(self):
        import time
        if self.gpu:
            self.e1.record()
        else:
            self.ms = 1e3 * (time.perf_counter() - self.t0)

    def elapsed(self):
        if self.gpu:
            self.e1.synchronize()
            return self.e0.elapsed_time(self.e1)
        return self.ms


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
