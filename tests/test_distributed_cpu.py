"""world_size-2 gloo test of the clip-sharded sequence path (stemseg_amd.pipeline.run_sequence_sharded):
clips dealt round-robin, one all-gather of per-clip head outputs, replicated chain -> every rank must reproduce the
single-process (reference-generated) golden result bit for bit.  Device ops are the oracle twin (no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import GOLDEN, ROOT


def _worker(rank, world, port, tag, q):
    for p in (ROOT, os.path.join(ROOT, "stem-seg_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stemseg_amd import config
        from stemseg_amd.inference.clusterers import SequentialClustering
        from stemseg_amd.inference.online_chainer import OnlineChainer
        from stemseg_amd.pipeline import run_sequence_sharded, shard_clips
        from tests.oracle_ops import OracleChainerOps
        g = np.load(os.path.join(GOLDEN, "chainer.npz"))
        emb, bw, sd, fg = g[tag + "__emb"], g[tag + "__bw"], g[tag + "__sd"], g[tag + "__fg"]
        clips = g[tag + "__subseqs"].tolist()
        overlap = len(set(clips[0]) & set(clips[1])) if len(clips) > 1 else 4
        config.load_preset("davis")
        calls = []

        def embed(frames):
            calls.append(list(frames))
            return (torch.from_numpy(emb[:, frames].copy()), torch.from_numpy(bw[:, frames].copy()), torch.from_numpy(sd[:, frames].copy()))
        chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0, ops=OracleChainerOps())
        (track, counts, life), _, _, _, meta = run_sequence_sharded(
            fg.shape[0], embed, chainer, "davis", frame_overlap=overlap, fg_mask_fn=lambda entries, thr: torch.from_numpy(fg))
        ok = all(np.array_equal(l.numpy(), g["%s_track_%02d" % (tag, t)]) for t, l in enumerate(track))
        ok = ok and sorted(counts.items()) == [tuple(r) for r in g[tag + "__pt_counts"].tolist()]
        ok = ok and all(meta[i]["instance_labels"] == g["%s_clip%d_instance_labels" % (tag, i)].tolist() for i in range(len(clips)))
        ok = ok and calls == [clips[i] for i in shard_clips(len(clips), rank, world)]     # each rank embedded only its own clips
        q.put((rank, bool(ok), len(calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["seq20_ov4", "seq14_ov6", "seq8_single"])
def test_sharded_sequence_two_ranks_gloo(tag):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + {"seq20_ov4": 0, "seq14_ov6": 1, "seq8_single": 2}[tag]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tag, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    n_clips = {"seq20_ov4": 4, "seq14_ov6": 4, "seq8_single": 1}[tag]
    assert sum(r[2] for r in res) == n_clips


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
