"""gloo tests (world 1 / 2 / 3 / 8) of the clip-parallel sequence path (stemseg_amd.pipeline.run_sequence_sharded):
every rank embeds AND clusters only its own clips with label_start = 1, two all-gathers (seediness planes; one-byte label
codes + clustering records), the Hungarian chain runs on label-pair tables -> every rank must reproduce the single-process
(reference-generated) golden result bit for bit.  Device ops are the oracle twin (no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import GOLDEN, ROOT


def _case(tag):
    """-> (emb, bw, sd, fg, clips, overlap, expected dict) for a ``chainer.npz`` tag or the 48-clip ``long`` sequence."""
    from tests import synth
    if tag == "long":
        g = np.load(os.path.join(GOLDEN, "chainer_long.npz"))
        emb, bw, sd, fg = synth.synth_long_sequence(int(g["n_clips"]), seed=int(g["seed"]))
        _, _, exp = synth.long_sequence_case(g, torch.from_numpy)
        clips = [list(range(4 * i, 4 * i + 8)) for i in range(int(g["n_clips"]))]
        return emb, bw, sd, fg, clips, 4, exp
    if tag == "ties":                                   # head outputs differ PER CLIP on the shared frames: keyed by the clip's frames
        g = np.load(os.path.join(GOLDEN, "chainer_ties.npz"))
        per_clip, fg, clips = synth.synth_tie_sequence(int(g["n_clips"]), seed=int(g["seed"]))
        _, _, exp = synth.tie_sequence_case(g, torch.from_numpy)
        by_clip = {tuple(fr): pc for fr, pc in zip(clips, per_clip)}
        return by_clip, None, None, fg, clips, 4, exp
    g = np.load(os.path.join(GOLDEN, "chainer.npz"))
    emb, bw, sd, fg = g[tag + "__emb"], g[tag + "__bw"], g[tag + "__sd"], g[tag + "__fg"]
    clips = g[tag + "__subseqs"].tolist()
    exp = dict(track=[g["%s_track_%02d" % (tag, t)] for t in range(fg.shape[0])],
               counts=[tuple(r) for r in g[tag + "__pt_counts"].tolist()], life=[tuple(r) for r in g[tag + "__lifetimes"].tolist()],
               instance_labels=[g["%s_clip%d_instance_labels" % (tag, i)].tolist() for i in range(len(clips))],
               clip_labels=[g["%s_clip%d_labels" % (tag, i)] for i in range(len(clips))])
    overlap = len(set(clips[0]) & set(clips[1])) if len(clips) > 1 else 4
    return emb, bw, sd, fg, clips, overlap, exp


def _run(tag, fn_name="run_sequence_sharded", comm=None):
    """The sharded driver on this process's rank; -> (ok, clips embedded here, clips clustered here, expected own clips)."""
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from tests.oracle_ops import OracleChainerOps
    emb, bw, sd, fg, clips, overlap, exp = _case(tag)
    config.load_preset("davis")
    calls, clustered = [], []

    def embed(frames):
        calls.append(list(frames))
        if isinstance(emb, dict):
            return tuple(torch.from_numpy(a.copy()) for a in emb[tuple(frames)])
        return (torch.from_numpy(emb[:, frames].copy()), torch.from_numpy(bw[:, frames].copy()), torch.from_numpy(sd[:, frames].copy()))

    class CountingOps(OracleChainerOps):
        def cluster(self, clusterer, pts, label_start, want_masks):
            clustered.append(label_start)
            return super().cluster(clusterer, pts, label_start, want_masks)
    chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0, ops=CountingOps())
    stats = {}
    (track, counts, life), mask_idxes, clip_labels, _, meta = getattr(pipeline, fn_name)(
        fg.shape[0], embed, chainer, "davis", frame_overlap=overlap, fg_mask_fn=lambda entries, thr: torch.from_numpy(fg), stats=stats,
        **({"comm": comm} if comm is not None else {}))
    ok = len(track) == len(exp["track"]) and all(l.dtype == torch.int64 and np.array_equal(l.numpy(), e) for l, e in zip(track, exp["track"]))
    ok = ok and sorted(counts.items()) == exp["counts"] and sorted(life.items()) == exp["life"]
    ok = ok and [m["instance_labels"] for m in meta] == exp["instance_labels"]
    for t in range(fg.shape[0]):
        ys, xs = np.nonzero(fg[t])
        ok = ok and np.array_equal(mask_idxes[t][0].numpy(), ys) and np.array_equal(mask_idxes[t][1].numpy(), xs)
    if "clip_labels" in exp:
        ok = ok and all(np.array_equal(torch.cat(clip_labels[i]).numpy(), exp["clip_labels"][i]) for i in range(len(clips)))
    rank = comm.rank if comm is not None else (dist.get_rank() if dist.is_initialized() else 0)
    world = comm.world if comm is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    mine = [clips[i] for i in pipeline.shard_clips(len(clips), rank, world)]
    return bool(ok), calls, clustered, mine, len(clips)


def _worker(rank, world, port, tag, fn_name, q):
    for p in (ROOT, os.path.join(ROOT, "stem-seg_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok, calls, clustered, mine, n_clips = _run(tag, fn_name)
        ok = ok and calls == mine                                         # each rank embedded only its own clips
        ok = ok and clustered == [1] * len(mine)                          # ... and clustered only those, with label_start = 1
        q.put((rank, bool(ok), len(calls), len(clustered)))
    finally:
        dist.destroy_process_group()


_PORTS = {}


def _spawn(world, tag, fn_name="run_sequence_sharded"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 300) + len(_PORTS)
    _PORTS[(world, tag, fn_name)] = port
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True] * world, res
    return res


@pytest.mark.parametrize("tag", ["seq20_ov4", "seq14_ov6", "seq8_single", "long", "ties"])
def test_clip_parallel_chain_one_process_vs_golden(tag):
    """world 1 (no process group): the table-based chain alone against the reference's outputs -- tracks, per-clip label lists,
    coordinates, counts, lifetimes, instance lists; `long` drives the track ids to 527; `ties` has exact Hungarian cost ties whose
    outcome depends on the reference's id enumeration order (online_chainer.reference_id_order)."""
    ok, calls, clustered, mine, n_clips = _run(tag)
    assert ok and len(calls) == n_clips and clustered == [1] * n_clips


@pytest.mark.parametrize("tag", ["seq20_ov4", "seq14_ov6", "seq8_single"])
def test_sharded_sequence_two_ranks_gloo(tag):
    res = _spawn(2, tag)
    n_clips = {"seq20_ov4": 4, "seq14_ov6": 4, "seq8_single": 1}[tag]
    assert sum(r[2] for r in res) == n_clips and sum(r[3] for r in res) == n_clips


@pytest.mark.parametrize("world,tag", [(3, "seq20_ov4"), (3, "seq14_ov6"), (8, "long"), (8, "seq20_ov4"), (3, "ties")])
def test_sharded_sequence_three_and_eight_ranks_gloo(world, tag):
    """world 3: uneven blocks; world 8 on the 48-clip sequence: 6 clips per rank, ids to 527; world 8 on 4 clips: idle ranks."""
    res = _spawn(world, tag)
    n_clips = {"seq20_ov4": 4, "seq14_ov6": 4, "long": 48, "ties": 12}[tag]
    assert sum(r[2] for r in res) == n_clips and sum(r[3] for r in res) == n_clips


@pytest.mark.parametrize("world,tag", [(4, "seq14_ov6"), (5, "long")])
def test_virtual_ranks_in_one_process(world, tag):
    """The thread-based virtual-rank harness (tests/virtual_ranks.py) that the GPU suite uses on its single device."""
    from tests.virtual_ranks import run_virtual_ranks
    res = run_virtual_ranks(world, lambda comm: _run(tag, comm=comm))
    assert all(r[0] for r in res)
    assert [r[1] for r in res] == [r[3] for r in res] and all(r[2] == [1] * len(r[3]) for r in res)


def test_world_1_process_group_runs_both_collectives_gloo():
    """A process group of ONE rank still runs the two exchanges (what tests/test_gpu_nccl.py does with RCCL on the GPU box)."""
    res = _spawn(1, "seq14_ov6")
    assert res[0][2] == 4 and res[0][3] == 4


def test_sharded_path_refuses_pre_resized_seediness_with_a_clear_message():
    """a model with a SEPARATE seediness head resizes its seediness itself under --resize_embeddings
    (inference_model.py:156); the chainer would resize it again (online_chainer.py:127-140) and the reference fails on the size
    mismatch.  The sharded driver says so instead of failing in a reshape (or resizing twice)."""
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from tests.oracle_ops import OracleChainerOps
    emb, bw, sd, fg, clips, overlap, _ = _case("seq20_ov4")
    config.load_preset("davis")
    chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 2.0, ops=OracleChainerOps())

    def embed(fr):                                          # seediness already x2, as InferenceModel._run_heads hands it out
        s2 = torch.from_numpy(sd[:, fr].copy()).repeat_interleave(2, -1).repeat_interleave(2, -2)
        return torch.from_numpy(emb[:, fr].copy()), torch.from_numpy(bw[:, fr].copy()), s2
    with pytest.raises(ValueError, match="separate seediness head"):
        pipeline.run_sequence_sharded(fg.shape[0], embed, chainer, "davis", frame_overlap=overlap)


def test_shard_clips_contiguous_blocks():
    """Each rank takes a contiguous, balanced block (neighbouring clips share frames: one trunk pass per shared frame)."""
    from stemseg_amd.pipeline import clip_owner, shard_clips
    assert shard_clips(15, 0, 8) == [0, 1] and shard_clips(15, 6, 8) == [12, 13] and shard_clips(15, 7, 8) == [14]
    assert shard_clips(8, 3, 8) == [3] and shard_clips(3, 5, 8) == [] and shard_clips(3, 2, 8) == [2]
    for n in (1, 5, 8, 9, 15, 16, 29):
        for w in (1, 2, 3, 8):
            blocks = [shard_clips(n, r, w) for r in range(w)]
            assert sum(blocks, []) == list(range(n)) and max(map(len, blocks)) - min(map(len, blocks)) <= 1
            assert all(clip_owner(ci, n, w) == (r, k) for r, b in enumerate(blocks) for k, ci in enumerate(b))


def test_sharded_sequence_refuses_a_non_finite_head_output_on_every_rank():
    """Overflow guard of the clip-parallel path: the owner of a clip whose head outputs hold NaN flags it in the byte that travels
    with the clustering record, and EVERY rank raises (none stitches NaN-derived labels)."""
    from stemseg_amd import config, hip, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from tests.oracle_ops import OracleChainerOps
    from tests.virtual_ranks import run_virtual_ranks
    emb, bw, sd, fg, clips, overlap, _ = _case("seq20_ov4")
    emb = emb.copy()
    emb[1, clips[2][3], 2, 5] = np.nan                      # one voxel of a frame only clip 2 (and its neighbours) holds
    config.load_preset("davis")

    def one(comm):
        chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0, ops=OracleChainerOps())
        embed = lambda fr: (torch.from_numpy(emb[:, fr].copy()), torch.from_numpy(bw[:, fr].copy()), torch.from_numpy(sd[:, fr].copy()))
        try:
            pipeline.run_sequence_sharded(fg.shape[0], embed, chainer, "davis", frame_overlap=overlap, fg_mask_fn=lambda e, t: torch.from_numpy(fg), comm=comm)
        except hip.NonFiniteError:
            return True
        return False
    assert all(run_virtual_ranks(3, one))


def _semseg_case(r, binary):
    """A sequence for the semseg / resize variant of the clip-parallel path: the seq20_ov4 golden's maps serve as the LOW-resolution
    head outputs, every clip additionally carries foreground logits [Cfg, T, h, w] (different per clip on shared frames, so the
    cross-clip mean matters), and everything is resized x r before masks and clustering (--resize_embeddings)."""
    emb, bw, sd, fg, clips, overlap, _ = _case("seq20_ov4")
    rs = np.random.RandomState(11)
    fgl = {}
    for ci, fr in enumerate(clips):
        base = np.where(fg[fr] > 0, 2.0, -2.0).astype(np.float32) + rs.standard_normal(fg[fr].shape).astype(np.float32)
        fgl[ci] = np.stack([-base, base], 0) if binary else base[None]
    return emb, bw, sd, clips, overlap, fgl


def _run_semseg(r, binary, comm=None):
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from tests.oracle_ops import OracleChainerOps
    emb, bw, sd, clips, overlap, fgl = _semseg_case(r, binary)
    n_frames = emb.shape[1]
    config.load_preset("davis")
    ops = OracleChainerOps()
    # single process: the reference's order -- resize each clip's logits, accumulate per frame, mean, sigmoid / softmax, > 0.5; chain with resize
    fg_full = ops.fg_from_semseg([(list(fr), torch.from_numpy(fgl[ci])) for ci, fr in enumerate(clips)], n_frames, float(r))
    dicts = [dict(frames=list(fr), embeddings=torch.from_numpy(emb[:, fr].copy()), bandwidths=torch.from_numpy(bw[:, fr].copy()),
                  seediness=torch.from_numpy(sd[:, fr].copy())) for fr in clips]
    want = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), float(r), ops=OracleChainerOps()).process(fg_full, dicts)
    by = {tuple(fr): ci for ci, fr in enumerate(clips)}

    def embed(frames):
        ci = by[tuple(frames)]
        return (torch.from_numpy(emb[:, frames].copy()), torch.from_numpy(bw[:, frames].copy()), torch.from_numpy(sd[:, frames].copy()),
                torch.from_numpy(fgl[ci].copy()))
    chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), float(r), ops=OracleChainerOps())
    got = pipeline.run_sequence_sharded(n_frames, embed, chainer, "davis", frame_overlap=overlap, fg_logit_channels=2 if binary else 1,
                                        **({"comm": comm} if comm is not None else {}))
    (t0, c0, l0), m0, s0, _, meta0 = want
    (t1, c1, l1), m1, s1, _, meta1 = got
    ok = len(t0) == len(t1) and all(torch.equal(a, b) for a, b in zip(t0, t1)) and dict(c0) == dict(c1) and l0 == l1
    ok = ok and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(m0, m1))
    ok = ok and [m["instance_labels"] for m in meta0] == [m["instance_labels"] for m in meta1]
    ok = ok and all(torch.equal(torch.cat(a), torch.cat(b)) for a, b in zip(s0, s1))
    return bool(ok), int(fg_full.sum()), max(list(c0) + [0])


@pytest.mark.parametrize("r,binary", [(2, False), (2, True), (1, False), (4, False)])
def test_clip_parallel_chain_with_semseg_foreground_and_resize(r, binary):
    """The clip-parallel path for presets with a semseg head and --resize_embeddings: foreground from the
    exchanged foreground logits (1 channel of a multi-class head / both of a binary head), clustering at full resolution -- the
    single-process chain's result bit for bit, at world 1 and on three virtual ranks."""
    from tests.virtual_ranks import run_virtual_ranks
    ok, n_fg, top = _run_semseg(r, binary)
    assert ok and n_fg > 100 and top >= 3
    assert all(x[0] for x in run_virtual_ranks(3, lambda comm: _run_semseg(r, binary, comm=comm)))


def test_sharded_semseg_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 300) + 77
    procs = [ctx.Process(target=_semseg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res


def _semseg_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "stem-seg_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run_semseg(2, False)[0]))
    finally:
        dist.destroy_process_group()
