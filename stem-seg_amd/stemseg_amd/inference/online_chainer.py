"""Per-clip clustering + cross-clip tracklet stitching.

Counterpart of ``stemseg.inference.online_chainer`` (masks_to_coord_list :11-22, TrackContainer :25-117,
OnlineChainer :120-343) with the same call signatures and return structure.  What changed underneath:
  * head outputs stay on the GPU (the reference moves every clip D2H in inference_model.py:161-162 and H2D again here);
  * the per-frame nonzero / permute / boolean-index / cat chain (:258-281) is one compaction (stemseg_hip_fg_gather);
  * clustering is the multi-launch HIP clusterer with a single read-back;
  * the K1 x K2 (2 reductions + 2 ``.item()``) association loop (:318-328) is one histogram kernel
    (stemseg_hip_overlap_counts); the Hungarian step stays on the host (scipy), as in the reference;
  * relabelling (:219-224) is one LUT kernel.
Integer bookkeeping (label numbering, ``next_track_label = highest id + 1``, "previous clip only" overlap,
no IoU threshold) follows SURVEY.md Appendix A.3 exactly.
"""
from collections import defaultdict

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .. import hip


def masks_to_coord_list(masks):
    """masks [T,H,W] -> list over frames of (ys, xs) LongTensors, row-major order (online_chainer.py:11-22)."""
    out = []
    for t in range(masks.shape[0]):
        idx = torch.nonzero(masks[t], as_tuple=False)
        out.append((idx[:, 0], idx[:, 1]))
    return out


class TrackContainer(object):
    """Final stitched labels per frame + the highest id seen (online_chainer.py:25-117)."""

    def __init__(self, num_frames):
        self._frame_labels = [None] * num_frames
        self._highest_instance_id = 0

    def add_labels(self, frame_nums, labels, max_label=None):
        """labels: list of int64 tensors.  ``max_label`` (int) lets the caller supply max over the added labels
        when it already has it on the host (avoids one device sync per frame)."""
        assert all(self._frame_labels[t] is None for t in frame_nums)
        for t, lab in zip(frame_nums, labels):
            self._frame_labels[t] = lab
            if max_label is None and lab.numel() > 0:
                self._highest_instance_id = max(self._highest_instance_id, int(lab.max().item()))
        if max_label is not None:
            self._highest_instance_id = max(self._highest_instance_id, int(max_label))
        return self._highest_instance_id + 1

    def labels_exist(self, frame_num):
        return self._frame_labels[frame_num] is not None

    def has_fg_pixels(self, frame_num):
        assert self.labels_exist(frame_num)
        return self._frame_labels[frame_num].numel() > 0

    def get_labels(self, frame_nums):
        assert all(self.labels_exist(t) for t in frame_nums)
        return [self._frame_labels[t] for t in frame_nums]

    def get_track_mask_idxes(self):
        """-> (labels per frame [CPU], {id: #pixels}, {id: last_frame - first_frame}); -1 is a key when outliers exist."""
        counts, first, last = defaultdict(lambda: 0), {}, {}
        frame_labels = [l.cpu() for l in self._frame_labels]
        for t, lab in enumerate(frame_labels):
            ids, n = np.unique(lab.numpy(), return_counts=True)
            for i, c in zip(ids.tolist(), n.tolist()):
                counts[i] += c
                first[i] = min(first.get(i, 10000), t)
                last[i] = max(last.get(i, -1), t)
        return frame_labels, counts, {k: last[k] - first[k] for k in first}


class HipChainerOps(object):
    """Device operations of the chainer on libstemseg_hip.so (tests may inject an oracle-backed twin).  Tensors stay on the
    device they arrive on (the model's); host tensors go to ``device`` (TrackGenerator's ``clustering_device``)."""

    def __init__(self, device=None):
        self._offs_pinned = {}
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None and torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())

    def to_device(self, t):
        if not torch.is_tensor(t) or t.is_cuda:
            return t
        return t.to(self.device, non_blocking=True)

    def resize(self, x, scale):
        return hip.upsample_trilinear(x.contiguous().float(), 1, _int_scale(scale), _int_scale(scale))

    def gather(self, emb, bw, seed, fg):
        e, b, s, vox, offs = hip.fg_gather(emb.contiguous(), bw.contiguous(), seed.contiguous(), fg.contiguous())
        return dict(emb=e, bw=b, seed=s, vox=vox, offs=offs, T=fg.shape[0])

    def cluster(self, clusterer, pts, label_start, want_masks):
        labels, meta_dev, masks, _ = clusterer.enqueue(pts["emb"], pts["bw"], pts["seed"], label_start,
                                                       pts["offs"][pts["T"]:], want_masks)
        return labels, meta_dev, masks

    def read(self, pts, meta_dev):
        """Frame offsets + clustering record with ONE host synchronisation (both copies are enqueued, the record's read waits)."""
        key = (pts["offs"].device.index, pts["offs"].numel())
        buf = self._offs_pinned.get(key)
        if buf is None:
            buf = self._offs_pinned[key] = torch.empty(pts["offs"].numel(), dtype=torch.int64, pin_memory=True)
        buf.copy_(pts["offs"], non_blocking=True)
        meta = hip.read_cluster_meta(meta_dev)            # synchronises the stream
        return buf.tolist(), meta

    def label_sets(self, groups, cap):
        """groups: lists of (device, int64) label arrays -> (for each group the ascending ids > 0 that occur, for each group whether
        a negative (outlier) label occurs), with ONE read-back for all of them (the reference's unique() per group,
        online_chainer.py:304-308, and the highest id of :43-49)."""
        rows = []
        dev = next((l.device for g in groups for l in g), self.device)
        for g in groups:
            ls = [l.contiguous() for l in g if l.numel() > 0]
            if ls:
                present, mx = hip.label_presence(ls, cap)             # present: cap bytes + the "negative label seen" byte
                rows.append(torch.cat([present.to(torch.int64), mx]))
            else:
                rows.append(torch.zeros(cap + 2, dtype=torch.int64, device=dev))
        host = torch.stack(rows).cpu()                    # the one read-back
        out, neg = [], []
        for r in host:
            assert int(r[-1]) <= cap, "label %d beyond the stated bound %d" % (int(r[-1]) - 1, cap)
            out.append([i for i in torch.nonzero(r[:cap]).flatten().tolist() if i > 0])
            neg.append(bool(r[cap]))
        return out, neg

    def present_ids(self, labels_list, cap=None, with_outlier=False):
        """Ascending ids > 0 that occur in the (device, int64) label arrays -- the reference's ``unique()`` minus the outlier
        id (online_chainer.py:304-308).  ``cap``: an exclusive upper bound on the ids when the caller has one (labels are
        always below next_track_label + max_instances); without it one extra pass finds the maximum first.
        ``with_outlier``: -> (ids, whether a negative label occurs)."""
        ls = [l.contiguous() for l in labels_list]
        if not ls:
            return ([], False) if with_outlier else []
        if cap is None:
            _, mx = hip.label_presence(ls, 0)
            cap = int(mx.item())
        present, mx = hip.label_presence(ls, cap)
        host = torch.cat([present.to(torch.int64), mx]).cpu()           # one read-back
        assert int(host[-1]) <= cap, "label %d beyond the stated bound %d" % (int(host[-1]) - 1, cap)
        ids = [i for i in torch.nonzero(host[:cap]).flatten().tolist() if i > 0]
        return (ids, bool(host[cap])) if with_outlier else ids

    def overlap_counts(self, la, lb, ids_a, ids_b):
        """ids_*: candidate ids (> 0) in any order.  -> inter [Ka,Kb], cnt_a, cnt_b (rows / columns in the given order) as numpy int64."""
        def lut(ids):
            t = torch.full((max(ids) + 2 if ids else 1,), -1, dtype=torch.int32)
            if ids:
                t[torch.as_tensor(ids, dtype=torch.int64) + 1] = torch.arange(len(ids), dtype=torch.int32)
            return t.to(la.device)
        Ka, Kb = len(ids_a), len(ids_b)
        inter, ca, cb = hip.overlap_counts(la.contiguous(), lb.contiguous(), lut(ids_a), lut(ids_b), Ka, Kb)
        host = torch.cat([inter.reshape(-1), ca, cb]).cpu().numpy()      # one read-back
        return host[:Ka * Kb].reshape(Ka, Kb), host[Ka * Kb:Ka * Kb + Ka], host[Ka * Kb + Ka:]

    def relabel(self, labels, mapping):
        """mapping: dict old -> new.  In place."""
        hi = max(mapping) + 2
        m = torch.arange(-1, hi - 1, dtype=torch.int64)
        for o, n in mapping.items():
            m[o + 1] = n
        hip.relabel(labels, m.to(labels.device))
        return labels

    def max_label(self, labels_list):
        nz = [l.contiguous() for l in labels_list if l.numel() > 0]
        if not nz:
            return None
        _, mx = hip.label_presence(nz, 0)
        m = int(mx.item()) - 1
        return m if m >= 0 else -1            # all outliers: the reference's lab.max() is -1, max(0, -1) keeps 0

    # ---- clip-parallel stitching (pipeline.run_sequence_sharded): clip-local label codes, pair tables, LUT gather --------
    META_BYTES = None

    def compact(self, fg):
        """fg uint8 [F,h,w] -> (voxel_index int32 [V], frame_offsets int64 [F+1]) on the device (masks_to_coord_list of the
        whole sequence in one compaction; no sync)."""
        return hip.fg_compact(fg.contiguous())

    def codes_from_labels(self, pts, labels, label_start, out):
        """One byte per voxel of the clip (0 background, 1..K clip-local instance, 255 outlier) into ``out`` (uint8, contiguous)."""
        hip.labels_to_codes(labels, pts["vox"], pts["offs"][pts["T"]:], label_start, out)

    def meta_bytes(self):
        import ctypes
        return ctypes.sizeof(hip.ClusterMeta)

    def pack_meta(self, meta_dev):
        return meta_dev                                   # the device record itself (uint8 [sizeof(StemsegClusterMeta)])

    def unpack_meta(self, raw):
        return hip.ClusterMeta.from_buffer_copy(bytes(raw))

    def fg_from_semseg(self, per_clip, n_frames, resize_scale):
        """Foreground mask of a whole sequence from the clips' semseg foreground logits (inference_model.py:121-128: every slot of
        every clip, repeats included, adds its RESIZED logits to its frame and bumps the frame's count; :197-231: mean -> sigmoid of
        the foreground logit (1 channel given) or softmax channel 1 (2 channels); inference/main.py:142-144: > 0.5).
        per_clip: list of (frame numbers of the T slots, float32 [Cfg, T, h, w] on the device) in clip order -> uint8 [F, H, W]."""
        from ..modeling.inference_model import InferenceModel
        r = _int_scale(resize_scale)
        acc, counts = None, [0] * n_frames
        for frames, plane in per_clip:
            x = plane.contiguous().float()
            if r != 1:
                x = hip.upsample_trilinear(x, 1, r, r)
            if acc is None:
                acc = torch.zeros((n_frames, x.shape[0]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            InferenceModel._accumulate_semseg(acc, counts, x, list(frames))
        prob, _ = hip.semseg_masks(acc, torch.as_tensor(counts, dtype=torch.float32).to(acc.device), None)
        return hip.fg_mask(prob.contiguous(), 1.0, 0.5)

    def overflow_byte(self, block):
        """uint8 device scalar: 1 when the clip's head outputs (a tuple of tensors or one stacked block) hold inf / NaN; no sync."""
        ts = [t.contiguous() for t in (block if isinstance(block, (tuple, list)) else [block])]
        return hip.overflow_status(ts).any().to(torch.uint8)

    def pair_tables(self, codes, plane_a, plane_b, B):
        dev = codes.device
        return hip.pair_tables(codes, torch.as_tensor(plane_a, dtype=torch.int32).to(dev, non_blocking=True),
                               torch.as_tensor(plane_b, dtype=torch.int32).to(dev, non_blocking=True), B)

    def read_back(self, *tensors):
        """Device tensors -> numpy arrays with ONE host synchronisation (pinned staging buffers, reused per shape)."""
        bufs = []
        for k, t in enumerate(tensors):
            key = ("rb", k, t.device.index, t.dtype, tuple(t.shape))
            buf = self._offs_pinned.get(key)
            if buf is None:
                buf = self._offs_pinned[key] = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
            buf.copy_(t, non_blocking=True)
            bufs.append(buf)
        if tensors:
            torch.cuda.current_stream(tensors[0].device).synchronize()
        return [b.numpy().copy() for b in bufs]

    def labels_from_codes(self, codes, vox, items, lut, max_count, n_out):
        dev = codes.device
        return hip.codes_to_labels(codes, vox, torch.as_tensor(items, dtype=torch.int64).to(dev, non_blocking=True),
                                   torch.as_tensor(lut, dtype=torch.int64).to(dev, non_blocking=True), max_count, n_out)


def _int_scale(scale):
    s = int(round(float(scale)))
    if s < 1 or abs(float(scale) - s) > 1e-9:
        raise NotImplementedError("embedding resize factor %r: the HIP trilinear kernel takes positive integer scales (the "
                                  "reference only ever passes 1.0 or 4.0, inference/main.py:209-213)" % (scale,))
    return s


def reference_id_order(ids, has_outlier):
    """The ids of one side of the association in the order the REFERENCE enumerates them:
    ``list(set(labels.unique().tolist()) - {OUTLIER_LABEL})`` (online_chainer.py:308-309) -- CPython's set-iteration order of
    the set built from the ascending unique() list (the outlier id -1 included when it occurs), not ascending order:
    list(set([-1, 2, 9]) - {-1}) is [9, 2].  The order decides which pair the Hungarian solver returns on exact cost ties (every
    zero-IoU entry costs exactly 1.0 and every returned pair is accepted, :332-343), so it is part of the bookkeeping that has to
    match.  The same expression is evaluated here on the same values."""
    raw = ([-1] if has_outlier else []) + sorted(int(i) for i in ids)
    return list(set(raw) - {-1})


def association_from_counts(inter, ca, cb, ids_1, ids_2):
    """The Hungarian step of online_chainer.py:310-343 on the label-pair statistics: IoU costs in float32 exactly as the
    reference forms them (``1. - iou.item()`` stored into a float32 matrix, :327), every returned pair accepted."""
    I = np.asarray(inter).astype(np.float32).reshape(len(ids_1), len(ids_2))
    A = np.asarray(ca).astype(np.float32)[:, None]
    B = np.asarray(cb).astype(np.float32)[None, :]
    U = (A + B - I).astype(np.float32)
    iou = (I / U).astype(np.float32) if I.size else np.zeros_like(I)
    costs = (1. - iou.astype(np.float64)).astype(np.float32)
    recall = (I / A).astype(np.float32) if I.size else np.zeros_like(I)
    idx1, idx2 = linear_sum_assignment(costs)
    associations, un1, un2 = [], set(ids_1), set(ids_2)
    for i1, i2 in zip(idx1, idx2):
        associations.append((ids_1[i1], ids_2[i2]))
        un1.remove(ids_1[i1])
        un2.remove(ids_2[i2])
    return associations, un1, un2, costs[idx1, idx2], (recall, ids_1, ids_2)


def frame_sources(clip_frames):
    """Which (clip, slot) contributes each frame to the track container: clip 0 all of its frames, clip i the frames it does not
    share with clip i - 1 (online_chainer.py:196-229).  -> ({frame: (clip, slot)}, per clip the slots on the overlap with the
    previous clip, per clip the slots it adds)."""
    src, ov_js, new_js = {}, [], []
    prev = None
    for i, frames in enumerate(clip_frames):
        shared = set(frames).intersection(prev) if prev is not None else set()
        ov_js.append([j for j, t in enumerate(frames) if t in shared])
        new_js.append([j for j, t in enumerate(frames) if t not in shared])
        for j in new_js[-1]:
            assert frames[j] not in src, "frame %d would be added twice" % frames[j]
            src[frames[j]] = (i, j)
        for j in ov_js[-1]:
            assert frames[j] in src
        prev = frames
    return src, ov_js, new_js


def stitch_from_tables(clip_frames, tables, item_of, Ks, B):
    """The serial chain of OnlineChainer.process (online_chainer.py:193-236) on label-pair TABLES instead of label arrays --
    what is left on every rank when the clips were clustered elsewhere with label_start = 1.

    clip_frames: per clip its (distinct) frame numbers; tables: int [n_items, B, B], tables[item_of[(i, j)]][a][b] = voxels of
    clip i's slot j whose code in the SOURCE plane of that frame (frame_sources) bins to a and whose code in clip i's own plane
    bins to b (a = 0 when the frame has no source yet; bins: 1..B-2 clip-local instances, B-1 the outlier label);
    Ks: instances per clip.  -> dict(lut_track, lut_sub: {(i, j): int64 [B] final label per bin}, label_start, instance_labels
    per clip, next_track_label, src / new_js of frame_sources)."""
    src, ov_js, new_js = frame_sources(clip_frames)
    nb = B - 2
    next_label, highest = 1, 0
    lut_track, lut_sub, starts, inst = {}, {}, [], []
    for i, frames in enumerate(clip_frames):
        start = next_label
        starts.append(start)
        base = np.full(B, -1, np.int64)
        base[0] = 0
        base[1:B - 1] = start + np.arange(nb, dtype=np.int64)
        labels_i = [start + k for k in range(int(Ks[i]))]
        mapping = {}
        if ov_js[i]:
            tabs = np.stack([tables[item_of[(i, j)]] for j in ov_js[i]]).astype(np.int64)             # [J, B, B]
            fin = np.stack([lut_track[src[frames[j]]] for j in ov_js[i]])                             # [J, B] final id per source bin
            rows = tabs[:, 1:B - 1, :]                                                                 # source instances only
            row_ids = fin[:, 1:B - 1].reshape(-1)
            row_cnt = rows.sum(2).reshape(-1)
            keep = row_cnt > 0
            ids_1, inv = np.unique(row_ids[keep], return_inverse=True)
            inter_rows = np.zeros((len(ids_1), nb), np.int64)
            np.add.at(inter_rows, inv, rows[:, :, 1:B - 1].reshape(-1, nb)[keep])
            ca = np.zeros(len(ids_1), np.int64)
            np.add.at(ca, inv, row_cnt[keep])
            col_cnt = tabs.sum((0, 1))[1:B - 1]                                                        # over ALL source bins
            cols = np.flatnonzero(col_cnt > 0)
            ids_1 = [int(v) for v in ids_1]
            ids_2 = [int(base[1 + c]) for c in cols]
            if ids_1 or ids_2:
                assert not set(ids_1).intersection(ids_2), "Labels overlap: {}, {}".format(ids_1, ids_2)
                # rows / columns in the reference's enumeration order (reference_id_order); an outlier on the overlap frames is
                # bin B-1 of the source plane (rows) / of this clip's plane (columns)
                o1 = reference_id_order(ids_1, bool(tabs[:, B - 1, :].sum() > 0))
                o2 = reference_id_order(ids_2, bool(tabs[:, :, B - 1].sum() > 0))
                r_idx = [ids_1.index(v) for v in o1]
                c_idx = [ids_2.index(v) for v in o2]
                sub = inter_rows[:, cols][r_idx][:, c_idx] if (r_idx and c_idx) else np.zeros((len(r_idx), len(c_idx)), np.int64)
                associations = association_from_counts(sub, ca[r_idx], col_cnt[cols][c_idx], o1, o2)[0]
                mapping = {cur: assoc for assoc, cur in associations}
        final = base.copy()
        for cur, assoc in mapping.items():
            final[cur - start + 1] = assoc
        for j in ov_js[i]:
            lut_sub[(i, j)] = base
        for j in new_js[i]:
            lut_sub[(i, j)] = final if mapping else base
            lut_track[(i, j)] = lut_sub[(i, j)]
        if new_js[i]:
            cnt_new = np.sum([tables[item_of[(i, j)]].sum(0) for j in new_js[i]], 0).astype(np.int64)  # per bin of this clip
            present = np.flatnonzero(cnt_new[1:B - 1] > 0)
            if present.size:
                max_label = int(lut_sub[(i, new_js[i][0])][1 + present].max())
            else:
                max_label = -1 if int(cnt_new.sum()) > 0 else None
            if max_label is not None:
                highest = max(highest, max_label)
            next_label = highest + 1
        for cur, assoc in mapping.items():
            labels_i[labels_i.index(cur)] = assoc
        inst.append(labels_i)
    return dict(lut_track=lut_track, lut_sub=lut_sub, label_start=starts, instance_labels=inst, next_track_label=next_label,
                src=src, new_js=new_js, ov_js=ov_js)


class OnlineChainer(object):
    OUTLIER_LABEL = -1

    def __init__(self, clusterer, embedding_resize_factor, ops=None):
        self.clusterer = clusterer
        self.resize_scale = embedding_resize_factor
        self.ops = ops if ops is not None else HipChainerOps()

    @torch.no_grad()
    def resize_tensors(self, subseq):
        """x4 trilinear of embeddings, seediness and (already activated) bandwidths (online_chainer.py:127-140)."""
        if self.resize_scale == 1.0:
            return
        for k in ("embeddings", "seediness", "bandwidths"):
            subseq[k] = self.ops.resize(subseq[k], self.resize_scale)

    @torch.no_grad()
    def process(self, masks, subsequences, return_fg_embeddings=False):
        ops = self.ops
        num_frames = masks.shape[0]
        masks = ops.to_device(masks)
        H, W = masks.shape[-2:]
        mask_idxes = [None] * num_frames
        subseq_labels_list, subseq_meta, fg_embeddings = [], [], []
        track = TrackContainer(num_frames)
        next_track_label = 1
        prev_frames = None

        for i, subseq in enumerate(subsequences):
            if isinstance(subseq['frames'], dict):
                subseq['frames'] = sorted(subseq['frames'].keys())
            frames = list(subseq['frames'])
            for k in ("embeddings", "bandwidths", "seediness"):
                subseq[k] = ops.to_device(subseq[k])
            self.resize_tensors(subseq)
            assert subseq['embeddings'].shape[-2:] == masks.shape[-2:], \
                "Size mismatch between embeddings {} and masks {}".format(subseq['embeddings'].shape, masks.shape)

            labels_per_frame, pts, meta_info = self.cluster_subsequence(
                masks[torch.as_tensor(frames, device=masks.device)], subseq['embeddings'], subseq['bandwidths'],
                subseq['seediness'], next_track_label, return_fg_embeddings)
            offs = pts["offs_host"]
            for j, t in enumerate(frames):
                if mask_idxes[t] is None:
                    local = pts["vox"][offs[j]:offs[j + 1]].long() - j * H * W
                    mask_idxes[t] = (torch.div(local, W, rounding_mode="floor"), local % W)
            subseq_labels_list.append(labels_per_frame)
            if return_fg_embeddings:
                fg_embeddings.append(pts["emb"][:offs[-1]].cpu())

            id_bound = next_track_label + self.clusterer.max_instances        # every label of this clip is below it
            if i == 0:
                (ids_new,), _ = ops.label_sets([labels_per_frame], id_bound)
                next_track_label = track.add_labels(frames, labels_per_frame, max_label=self._max_of(ids_new, labels_per_frame))
                subseq_meta.append(meta_info)
                prev_frames = frames
                subseq["embeddings"] = subseq["bandwidths"] = subseq["seediness"] = None
                continue

            overlap = sorted(set(frames).intersection(prev_frames))        # previous clip only (:201-202)
            existing = track.get_labels(overlap)
            current = [labels_per_frame[j] for j, t in enumerate(frames) if t in overlap]
            new_js = [j for j, t in enumerate(frames) if t not in overlap]
            new_labels = [labels_per_frame[j] for j in new_js]
            # one read-back: ids on the overlap frames (existing / current) and ids of the frames this clip adds
            (ids_1, ids_2, ids_new), (neg_1, neg_2, _) = ops.label_sets([existing, current, new_labels], id_bound)
            ids_1, ids_2 = reference_id_order(ids_1, neg_1), reference_id_order(ids_2, neg_2)      # (decides exact cost ties)
            associations = self._associate_ids(torch.cat(list(existing)), torch.cat(list(current)), ids_1, ids_2)[0] if ids_1 or ids_2 \
                else []
            mapping = {cur: assoc for assoc, cur in associations}
            if mapping and new_js:
                # the added frames' labels are consecutive slices of the clip's label array: one launch per run of frames
                run_start = prev = new_js[0]
                for j in new_js[1:] + [None]:
                    if j is None or j != prev + 1:
                        ops.relabel(pts["labels"][offs[run_start]:offs[prev + 1]], mapping)
                        run_start = j
                    prev = j
            if new_js:
                mapped = [mapping.get(k, k) for k in ids_new]
                next_track_label = track.add_labels([frames[j] for j in new_js], new_labels, max_label=self._max_of(mapped, new_labels))
            for assoc, cur in associations:
                meta_info['instance_labels'][meta_info['instance_labels'].index(cur)] = assoc
            subseq_meta.append(meta_info)
            prev_frames = frames
            subseq["embeddings"] = subseq["bandwidths"] = subseq["seediness"] = None

        mask_idxes = [(ys.cpu(), xs.cpu()) for ys, xs in mask_idxes]
        subseq_labels_list = [[l.cpu() for l in ls] for ls in subseq_labels_list]
        return track.get_track_mask_idxes(), mask_idxes, subseq_labels_list, fg_embeddings, subseq_meta

    def cluster_subsequence(self, fg_clip, embeddings, bandwidths, seediness, label_start, return_fg_embeddings):
        """fg_clip [T',H,W] uint8, embeddings [E,T',H,W], bandwidths [Ev,T',H,W], seediness [1,T',H,W]
        -> (labels split per frame, gathered points dict, clustering meta dict)."""
        assert fg_clip.shape[0] == embeddings.shape[1]
        ops = self.ops
        pts = ops.gather(embeddings, bandwidths, seediness, fg_clip)
        labels, meta_dev, masks = ops.cluster(self.clusterer, pts, label_start, return_fg_embeddings)
        offs, meta = ops.read(pts, meta_dev)
        pts["offs_host"] = offs
        n = offs[-1]
        info = self.clusterer.meta_to_dict(meta, embeddings.shape[0], label_start, masks, None, n)
        assert labels.numel() >= n
        pts["labels"] = labels
        per_frame = [labels[offs[j]:offs[j + 1]] for j in range(len(offs) - 1)]
        return per_frame, pts, info

    @staticmethod
    def _max_of(ids, labels_list):
        """highest id among the labels being added, from the ids known to occur: -1 when there are points but only outliers,
        None when there is no point at all (TrackContainer.add_labels, online_chainer.py:43-49)."""
        if ids:
            return max(ids)
        return -1 if any(l.numel() > 0 for l in labels_list) else None

    def associate_clusters(self, labels_1, labels_2, id_bound=None):
        """Hungarian matching on 1 - IoU over the overlap frames (online_chainer.py:291-343).  Every returned pair is
        accepted -- there is no IoU threshold.  Returns the reference's 5-tuple.  ``id_bound`` (optional): an exclusive upper
        bound on the label ids, saves the pass that would find it."""
        la = labels_1 if torch.is_tensor(labels_1) else torch.cat(list(labels_1))
        lb = labels_2 if torch.is_tensor(labels_2) else torch.cat(list(labels_2))
        assert la.shape == lb.shape, "Shape mismatch: {}, {}".format(la.shape, lb.shape)
        if la.numel() == 0:
            return [], set(), set(), np.zeros(0, np.float32), (np.zeros((0, 0), np.float32), [], [])
        # only the ids that occur on the overlap frames enter the statistics (the reference's unique(), :304-308): the table
        # is K1 x K2 <= a few hundred cells however large the track ids have grown
        ids_1, ids_2 = self.ops.present_ids([la], id_bound, with_outlier=True), self.ops.present_ids([lb], id_bound, with_outlier=True)
        return self._associate_ids(la, lb, reference_id_order(*ids_1), reference_id_order(*ids_2))

    def _associate_ids(self, la, lb, ids_1, ids_2):
        assert la.shape == lb.shape, "Shape mismatch: {}, {}".format(la.shape, lb.shape)
        assert not set(ids_1).intersection(ids_2), "Labels overlap: {}, {}".format(ids_1, ids_2)
        inter, ca, cb = self.ops.overlap_counts(la, lb, ids_1, ids_2)
        return association_from_counts(inter, ca, cb, ids_1, ids_2)
