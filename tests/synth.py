"""Deterministic synthetic weights / inputs shared by the golden generator (tools/make_goldens.py),
the oracle checks and the GPU parity tests.

Everything here is derived from numpy ``RandomState`` streams keyed by *name*, so the values do not
depend on torch's RNG, on construction order, or on which process (reference import vs. this repo's
own modules) asks for them.  Data only -- no reference code involved.
"""
import zlib

import numpy as np


def _rng(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_param(name, shape, seed=0):
    """Value for one state-dict entry.  Rules are keyed on the *suffix* of the reference key names
    (checkpoint row) so the same function serves backbone, FPN and decoders."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(name, seed)
    if len(shape) == 0:
        return np.float32(1.0)                       # e.g. embedding_head.time_scale
    n = rng.standard_normal(shape).astype(np.float32)
    if len(shape) >= 2:                              # conv weight [Cout, Cin, k...]: He-normal
        fan_in = int(np.prod(shape[1:]))
        return (n * np.float32(np.sqrt(2.0 / fan_in))).astype(np.float32)
    if name.endswith("running_var"):
        return (0.5 + 0.5 * np.abs(n)).astype(np.float32)
    if name.endswith("running_mean"):
        return (0.1 * n).astype(np.float32)
    if name.endswith("bn3.weight"):                  # keep the residual branch small -> no blow-up
        return (0.3 + 0.05 * n).astype(np.float32)
    if name.endswith(".weight"):                     # GroupNorm / FrozenBN scale
        return (1.0 + 0.2 * n).astype(np.float32)
    return (0.1 * n).astype(np.float32)              # every bias


def synth_state_dict(named_shapes, seed=0, prefix=""):
    """named_shapes: iterable of (key, shape).  Returns {key: np.float32 array}."""
    return {k: synth_param(prefix + k, s, seed) for k, s in named_shapes}


def synth_features(T, H32, W32, C=256, seed=0, dtype=np.float32):
    """Four FPN-like feature stacks [C, T, h, w] ordered 32x, 16x, 8x, 4x."""
    out = []
    for i, s in enumerate((1, 2, 4, 8)):
        rng = _rng("feat%d" % i, seed)
        out.append(rng.standard_normal((C, T, H32 * s, W32 * s)).astype(dtype))
    return out


def synth_frames(T, H, W, seed=0):
    """uint8 BGR frames [T, H, W, 3]."""
    return np.random.RandomState(seed).randint(0, 256, size=(T, H, W, 3)).astype(np.uint8)


def synth_cluster_case(T, H, W, K, E=4, Ev=2, seed=0, bg_fraction=0.15, noise=0.05,
                       free_dims=2, seed_peak=1.0):
    """Structured head outputs: K instances as moving boxes (SURVEY.md section 8(d) 'clustering driver').

    Returns emb [E,T,H,W], bw [Ev,T,H,W], seed [1,T,H,W] (float32) and fg [T,H,W] uint8.
    Instance centres are spread on a lattice so that clusters are well separated (margin cases);
    background pixels get far-away embeddings and low seediness.
    """
    rng = np.random.RandomState(1000 + seed)
    emb = (10.0 + 3.0 * rng.standard_normal((E, T, H, W))).astype(np.float32)
    bw = (25.0 + rng.uniform(0, 1, size=(Ev, T, H, W))).astype(np.float32)
    sd = rng.uniform(0, 0.2, size=(1, T, H, W)).astype(np.float32)
    fg = np.zeros((T, H, W), np.uint8)
    if K > 0:
        cols = int(np.ceil(np.sqrt(K)))
        bh, bw_ = max(2, H // (cols + 1)), max(2, W // (cols + 1))
        for k in range(K):
            cy, cx = (k // cols), (k % cols)
            y0 = int(cy * (H - bh) / max(cols - 1, 1))
            x0 = int(cx * (W - bw_) / max(cols - 1, 1))
            centre = np.array([-1.5 + 3.0 * (cy + 0.5) / cols, -1.5 + 3.0 * (cx + 0.5) / cols] +
                              [0.6 * ((k * 7) % 5 - 2) for _ in range(E - 2)], np.float32)[:E]
            for t in range(T):
                dx = min(t, max(W - (x0 + bw_), 0))      # box drifts right, clipped
                ys, xs = slice(y0, y0 + bh), slice(x0 + dx, x0 + dx + bw_)
                nz = (noise * rng.standard_normal((E, bh, bw_))).astype(np.float32)
                emb[:, t, ys, xs] = centre[:, None, None] + nz
                nrm = np.sqrt((nz ** 2).sum(0))
                sd[0, t, ys, xs] = np.clip(seed_peak - nrm, 0, 1)
                fg[t, ys, xs] = 1
    # sprinkle some background into the fg mask so outliers (-1) exist
    extra = rng.uniform(size=(T, H, W)) < bg_fraction * 0.1
    fg = np.where(extra, 1, fg).astype(np.uint8)
    return emb, bw, sd, fg


def synth_long_sequence(n_clips, H=24, W=48, births_per_clip=4, seed=0, T=8, overlap=4):
    """A long sequence whose TRACK IDS KEEP GROWING: every clip (T frames, stride T - overlap) sees ``births_per_clip`` new
    instances appear on its first non-overlap frame; each lives for T frames and then dies, so it is born unmatched in one
    clip, matched in the next two, and the sequence-wide id counter climbs by ~births_per_clip per clip (the regime of
    long KITTI-MOTS sequences).  Returns emb [4,F,H,W], bw [2,F,H,W], seed [1,F,H,W] float32 and fg [F,H,W] uint8."""
    stride = T - overlap
    F = stride * (n_clips - 1) + T
    rng = np.random.RandomState(7000 + seed)
    emb = (10 + 3 * rng.standard_normal((4, F, H, W))).astype(np.float32)
    bw = (25 + rng.uniform(0, 1, (2, F, H, W))).astype(np.float32)
    sd = rng.uniform(0, 0.2, (1, F, H, W)).astype(np.float32)
    fg = np.zeros((F, H, W), np.uint8)
    rows = 3
    ch, cw = H // rows, W // births_per_clip
    bh, bwid = max(2, ch - 3), max(2, cw - 4)
    k = 0
    for gen in range(-2, n_clips):                      # generations -2, -1 populate the first clip's overlap-free start
        first = stride * gen + overlap
        for j in range(births_per_clip):
            y0, x0 = (gen % rows) * ch + 1, j * cw + 2
            free = (0.6 * ((k * 7) % 5 - 2), 0.6 * ((k * 3) % 5 - 2))
            k += 1
            for t in range(max(first, 0), min(first + T, F)):
                c = np.array([-1 + 2 * (y0 + bh / 2) / H, -1.3 + 2.6 * (x0 + bwid / 2) / W, free[0], free[1]], np.float32)
                nz = (0.04 * rng.standard_normal((4, bh, bwid))).astype(np.float32)
                emb[:, t, y0:y0 + bh, x0:x0 + bwid] = c[:, None, None] + nz
                sd[0, t, y0:y0 + bh, x0:x0 + bwid] = np.clip(1 - np.sqrt((nz ** 2).sum(0)), 0, 1)
                fg[t, y0:y0 + bh, x0:x0 + bwid] = 1
    fg = np.where(rng.uniform(size=fg.shape) < 0.01, 1, fg).astype(np.uint8)
    return emb, bw, sd, fg


def synth_tie_sequence(n_clips, H=24, W=48, seed=0, T=8, overlap=4, p_coherent=0.7):
    """A sequence that makes the chainer's Hungarian step hit EXACT COST TIES on rectangular matrices: 18 persistent objects
    (3 x 6 cells); per CLIP each object is either coherent (tight embedding, high seediness -> one instance) or noise (scattered
    embedding, low seediness -> outlier points), drawn independently per clip.  An object that is an instance in clip i-1 and noise
    in clip i is an old id with zero IoU against everything; one that is noise in i-1 and an instance in i is a new id with zero
    IoU against everything: with two or more of the former, which old track the new instance is merged into depends on the ORDER
    the ids are enumerated in (online_chainer.py:308-309: list(set(unique) - {-1})).  Head outputs therefore differ per clip on
    the shared frames.  -> (per clip (emb [4,T,H,W], bw [2,T,H,W], seed [1,T,H,W]) float32, fg [F,H,W] uint8, clips)."""
    stride = T - overlap
    F = stride * (n_clips - 1) + T
    rng = np.random.RandomState(9100 + seed)
    rows, cols = 3, 6
    ch, cw = H // rows, W // cols
    bh, bwid = ch - 2, cw - 2
    fg = np.zeros((F, H, W), np.uint8)
    cells = [(r * ch + 1, c * cw + 1) for r in range(rows) for c in range(cols)]
    for (y0, x0) in cells:
        fg[:, y0:y0 + bh, x0:x0 + bwid] = 1
    clips = [list(range(stride * i, stride * i + T)) for i in range(n_clips)]
    per_clip = []
    for i in range(n_clips):
        emb = (10 + 3 * rng.standard_normal((4, T, H, W))).astype(np.float32)
        bw = (25 + rng.uniform(0, 1, (2, T, H, W))).astype(np.float32)
        sd = rng.uniform(0, 0.2, (1, T, H, W)).astype(np.float32)
        coherent = rng.uniform(size=len(cells)) < p_coherent
        for k, (y0, x0) in enumerate(cells):
            if not coherent[k]:
                continue
            free = (0.6 * ((k * 7) % 5 - 2), 0.6 * ((k * 3) % 5 - 2))
            c = np.array([-1 + 2 * (y0 + bh / 2) / H, -1.3 + 2.6 * (x0 + bwid / 2) / W, free[0], free[1]], np.float32)
            nz = (0.04 * rng.standard_normal((4, T, bh, bwid))).astype(np.float32)
            emb[:, :, y0:y0 + bh, x0:x0 + bwid] = c[:, None, None, None] + nz
            sd[0, :, y0:y0 + bh, x0:x0 + bwid] = np.clip(1 - np.sqrt((nz ** 2).sum(0)), 0, 1)
        per_clip.append((emb, bw, sd))
    return per_clip, fg, clips


def tie_sequence_case(golden_npz, to_tensor):
    """(fg, clip dicts, expected) of the ``chainer_ties`` golden (inputs regenerated from the seed, checksums verified)."""
    import zlib
    g = golden_npz
    per_clip, fg, clips = synth_tie_sequence(int(g["n_clips"]), seed=int(g["seed"]))
    crc = [zlib.crc32(np.concatenate([a.reshape(-1) for a in pc]).tobytes()) for pc in per_clip] + [zlib.crc32(fg.tobytes())]
    assert crc == g["input_crc"].tolist(), "synthetic inputs drifted"
    dicts = [dict(frames=list(fr), embeddings=to_tensor(e.copy()), bandwidths=to_tensor(b.copy()), seediness=to_tensor(s.copy()))
             for fr, (e, b, s) in zip(clips, per_clip)]
    track = np.split(g["track_labels"].astype(np.int64), np.cumsum(g["track_sizes"])[:-1])
    inst = np.split(g["instance_labels"], np.cumsum(g["instance_label_sizes"])[:-1])
    exp = dict(track=track, counts=[tuple(r) for r in g["pt_counts"].tolist()], life=[tuple(r) for r in g["lifetimes"].tolist()],
               instance_labels=[a.tolist() for a in inst])
    return fg, dicts, exp


def long_sequence_case(golden_npz, to_tensor):
    """(fg, clip dicts, expected) of the ``chainer_long`` golden: inputs regenerated from the stored seed (checksums
    verified), expected outputs as the reference's chainer produced them."""
    import zlib
    g = golden_npz
    n_clips = int(g["n_clips"])
    emb, bw, sd, fg = synth_long_sequence(n_clips, seed=int(g["seed"]))
    assert [zlib.crc32(a.tobytes()) for a in (emb, bw, sd, fg)] == g["input_crc"].tolist(), "synthetic inputs drifted"
    F = fg.shape[0]
    clips = [list(range(4 * i, 4 * i + 8)) for i in range(n_clips)]
    assert clips[-1][-1] == F - 1
    dicts = [dict(frames=list(fr), embeddings=to_tensor(emb[:, fr].copy()), bandwidths=to_tensor(bw[:, fr].copy()),
                  seediness=to_tensor(sd[:, fr].copy())) for fr in clips]
    track = np.split(g["track_labels"].astype(np.int64), np.cumsum(g["track_sizes"])[:-1])
    inst = np.split(g["instance_labels"], np.cumsum(g["instance_label_sizes"])[:-1])
    exp = dict(track=track, counts=[tuple(r) for r in g["pt_counts"].tolist()], life=[tuple(r) for r in g["lifetimes"].tolist()],
               instance_labels=[a.tolist() for a in inst])
    return fg, dicts, exp


def check_long_sequence(result, exp):
    (track, counts, life), _, _, _, meta = result
    assert len(track) == len(exp["track"])
    for t, l in enumerate(track):
        assert np.array_equal(l.cpu().numpy(), exp["track"][t]), "frame %d: track labels differ from the reference" % t
    assert sorted(counts.items()) == exp["counts"] and sorted(life.items()) == exp["life"]
    assert [m["instance_labels"] for m in meta] == exp["instance_labels"]
    return max(counts)
