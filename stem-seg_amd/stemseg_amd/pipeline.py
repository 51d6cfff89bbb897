"""Clip-level hot path (embed + cluster) and its multi-GPU sharding.

* ``ClipPipeline.step`` is BASELINE.json's unit of work: one T-frame clip through encoder -> two 3-D decoders ->
  fused heads -> fg mask -> fg gather -> sequential clustering, all enqueued on one HIP stream with no host
  synchronisation (N, K and the instance list stay on the device until the caller reads them).
* ``run_sequence_sharded`` is the one-process-per-GPU form for long sequences: clips are dealt round-robin to
  ranks, every rank embeds its own clips, ONE all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests)
  exchanges the per-clip head outputs (<= 5.8 MB per clip at 480p), and the cheap chain (fg mask from the
  cross-clip mean seediness, clustering, Hungarian stitching: < 1 % of the work) is replicated so every rank
  ends with the single-process result bit for bit.
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

    @torch.no_grad()
    def embed(self, frames):
        """frames: float32 [T,3,H,W] on the device (pre-processed, H and W multiples of 32)."""
        return self.model.embed_frames(frames.contiguous())

    @torch.no_grad()
    def cluster(self, emb, bw, seed, label_start=1):
        """Single-clip fg mask (seediness > thr), gather, clustering.  Returns device tensors only."""
        T = seed.shape[1]
        acc = torch.empty_like(seed[0])
        hip.seediness_accumulate(acc, seed[0].contiguous(), True)
        fg = hip.fg_mask(acc, 1.0, self.seediness_thresh)
        e, b, s, vox, offs = hip.fg_gather(emb.contiguous(), bw.contiguous(), seed.contiguous(), fg)
        labels, meta_dev, _, _ = self.clusterer.enqueue(e, b, s, label_start, offs[T:])
        return dict(labels=labels, meta=meta_dev, voxel_index=vox, frame_offsets=offs, fg=fg)

    @torch.no_grad()
    def step(self, frames):
        emb, bw, seed = self.embed(frames)
        out = self.cluster(emb, bw, seed)
        out.update(emb=emb, bw=bw, seed=seed)
        return out


    @torch.no_grad()
    def step_batch(self, frames, n_clips):
        """``n_clips`` clips stacked along the frame axis ([n_clips * T, 3, H, W]): one encoder pass, then decoders + fg mask +
        gather + clustering per clip.  Returns a list of ``step``-style dicts."""
        outs = []
        for emb, bw, seed in self.model.embed_frames_batch(frames.contiguous(), n_clips):
            out = self.cluster(emb, bw, seed)
            out.update(emb=emb, bw=bw, seed=seed)
            outs.append(out)
        return outs

    @torch.no_grad()
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
    def __init__(self, pipe, example_frames, overlap=False, n_clips=None, lane=0):
        self.pipe = pipe
        self.lane = lane
        self.stream = torch.cuda.Stream(device=example_frames.device)       # replays of this lane are ordered on this stream
        pipe.model.set_lane(lane)
        fn = pipe.step if n_clips is None else (lambda x: pipe.step_batch(x, n_clips))     # n_clips: ``run`` returns a list
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
        finally:
            pipe.model.overlap_decoders = prev
            pipe.model.set_lane(0)

    def run(self, frames):
        """Device-to-device copy of the clip into the graph's input, one graph launch; returns the static output dict."""
        self.static_in.copy_(frames, non_blocking=True)
        self.graph.replay()
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


# ------------------------------------------------------------------------------------------------ multi-GPU
def shard_clips(n_clips, rank, world_size):
    """Round-robin deal: clip i -> rank i % world_size."""
    return [i for i in range(n_clips) if i % world_size == rank]


@torch.no_grad()
def run_sequence_sharded(n_frames, embed_clip_fn, chainer, dataset_name="davis", frame_overlap=-1, seediness_thresh=0.25,
                         fg_mask_fn=None, group=None):
    """embed_clip_fn(frame_indices) -> (emb [E,T,h,w], bw [Ev,T,h,w], seed [1,T,h,w]) on this rank's device.
    Returns OnlineChainer.process(...) output, identical on every rank."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    clips, _ = get_subsequence_frames(n_frames, cfg.INPUT.NUM_FRAMES, dataset_name, frame_overlap)
    mine = shard_clips(len(clips), rank, world)
    per_rank = (len(clips) + world - 1) // world
    packed = None
    outs = {}
    for slot, ci in enumerate(mine):
        emb, bw, seed = embed_clip_fn(clips[ci])
        blk = torch.cat([emb, bw, seed], 0)                         # [E+Ev+1, T, h, w]
        if packed is None:
            packed = torch.zeros((per_rank,) + tuple(blk.shape), dtype=blk.dtype, device=blk.device)
        packed[slot] = blk
        outs[ci] = (emb.shape[0], bw.shape[0])
    if distributed and world > 1:
        # every rank owns >= 1 clip whenever len(clips) >= world; otherwise learn the block shape from rank 0
        shape = torch.tensor(list(packed.shape) if packed is not None else [0] * 5, dtype=torch.int64,
                             device=packed.device if packed is not None else _default_device())
        shapes = [torch.zeros_like(shape) for _ in range(world)]
        dist.all_gather(shapes, shape, group=group)
        full = next(s for s in shapes if int(s[0]) > 0).tolist()
        if packed is None:
            packed = torch.zeros(full, dtype=torch.float32, device=shape.device)
        gathered = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(gathered, packed.contiguous(), group=group)   # the one data-path collective
    else:
        gathered = [packed]
    E = Ev = None
    for v in outs.values():
        E, Ev = v
    if distributed and world > 1:
        dims = torch.tensor([E or 0, Ev or 0], dtype=torch.int64, device=packed.device)
        all_dims = [torch.zeros_like(dims) for _ in range(world)]
        dist.all_gather(all_dims, dims, group=group)
        E, Ev = next((int(d[0]), int(d[1])) for d in all_dims if int(d[0]) > 0)
    entries = []
    for ci, frames in enumerate(clips):
        blk = gathered[ci % world][ci // world]
        uniq = sorted(set(frames))
        if len(uniq) != len(frames):
            sel = torch.as_tensor([max(j for j, v in enumerate(frames) if v == t) for t in uniq], device=blk.device)
            blk = blk[:, sel]
        entries.append(EmbeddingMapEntry(uniq, blk[:E], blk[E:E + Ev], blk[E + Ev:]))
    fg_fn = fg_mask_fn if fg_mask_fn is not None else fg_masks_from_seediness
    fg = fg_fn(entries, seediness_thresh)
    dicts = [{"frames": e.subseq_frames, "embeddings": e.embeddings, "bandwidths": e.bandwidths, "seediness": e.seediness}
             for e in entries]
    return chainer.process(fg, dicts)


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
