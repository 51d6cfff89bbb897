"""CPU twin of stemseg_amd.inference.online_chainer.HipChainerOps built on the oracle (tests only).

Lets the chainer's HOST logic (windowing, label bookkeeping, Hungarian stitching, track container) be checked
against the reference-generated goldens on a machine without a GPU, and gives the gloo multi-process tests a
device-free data path.  The product never imports this file.
"""
import numpy as np
import torch

from oracle import pipeline as opipe
from oracle.clusterer import sequential_clustering


class _Meta(object):
    pass


class OracleChainerOps(object):
    device = "cpu"

    def to_device(self, t):
        return t

    def resize(self, x, scale):
        return torch.nn.functional.interpolate(x[None], scale_factor=(1.0, scale, scale), mode="trilinear", align_corners=False)[0]

    def gather(self, emb, bw, seed, fg):
        e, b, s, counts = opipe.gather_fg(emb.numpy(), bw.numpy(), seed.numpy(), fg.numpy())
        T, H, W = fg.shape
        vox = torch.from_numpy(np.flatnonzero(fg.numpy().reshape(-1)).astype(np.int32))
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return dict(emb=torch.from_numpy(e), bw=torch.from_numpy(b), seed=torch.from_numpy(s[:, 0]), vox=vox,
                    offs=torch.from_numpy(offs), T=T)

    def cluster(self, clusterer, pts, label_start, want_masks):
        labels, meta = sequential_clustering(pts["emb"].numpy(), pts["bw"].numpy(), pts["seed"].numpy(), label_start=label_start,
                                             primary=clusterer.primary_prob_thresh, secondary=clusterer.secondary_prob_thresh,
                                             min_seediness=clusterer.min_seediness_prob,
                                             free_dim_stds=clusterer.free_dim_stds[:clusterer.n_free_dims],
                                             max_instances=clusterer.max_instances, return_masks=want_masks)
        m = _Meta()
        m.K = len(meta["instance_labels"])
        E = pts["emb"].shape[1]
        m.centers = [c + [0.0] * (8 - E) for c in meta["instance_centers"]]
        # meta_to_dict recomputes std from the bandwidth; invert exactly what it will do
        m.bandwidths = [[1.0 / (s * s) if s > 0 else 0.0 for s in stds] + [1.0] * (8 - E) for stds in meta["instance_stds"]]
        masks = torch.from_numpy(np.stack(meta["instance_masks"]).astype(np.uint8)) if (want_masks and meta["instance_masks"]) else None
        return torch.from_numpy(labels), m, masks

    def read(self, pts, meta):
        return pts["offs"].tolist(), meta

    def present_ids(self, labels_list, cap=None, with_outlier=False):
        ids = np.unique(np.concatenate([l.numpy().reshape(-1) for l in labels_list])) if labels_list else np.zeros(0, np.int64)
        assert cap is None or ids.size == 0 or ids.max() < cap
        out = [int(i) for i in ids if i > 0]
        return (out, bool((ids < 0).any())) if with_outlier else out

    def label_sets(self, groups, cap):
        res = [self.present_ids([l for l in g if l.numel() > 0], cap, with_outlier=True) if any(l.numel() > 0 for l in g) else ([], False) for g in groups]
        return [r[0] for r in res], [r[1] for r in res]

    def overlap_counts(self, la, lb, ids_a, ids_b):
        la, lb = la.numpy(), lb.numpy()
        inter = np.array([[np.sum((la == a) & (lb == b)) for b in ids_b] for a in ids_a], np.int64).reshape(len(ids_a), len(ids_b))
        return inter, np.array([np.sum(la == a) for a in ids_a], np.int64), np.array([np.sum(lb == b) for b in ids_b], np.int64)

    def relabel(self, labels, mapping):
        src = labels.clone()
        for old, new in mapping.items():
            labels[src == old] = new
        return labels

    def max_label(self, labels_list):
        nz = [l for l in labels_list if l.numel() > 0]
        return int(torch.cat(nz).max().item()) if nz else None

    # ---- clip-parallel stitching twins (pipeline.run_sequence_sharded) ---------------------------------------------------
    def compact(self, fg):
        f = fg.numpy().reshape(fg.shape[0], -1)
        vox = np.flatnonzero(f.reshape(-1)).astype(np.int32)
        offs = np.concatenate([[0], np.cumsum(f.astype(bool).sum(1))]).astype(np.int64)
        out = np.zeros(f.size, np.int32)
        out[:vox.size] = vox
        return torch.from_numpy(out), torch.from_numpy(offs)

    def codes_from_labels(self, pts, labels, label_start, out):
        n = int(pts["offs"][pts["T"]])
        lab = labels.numpy()[:n]
        out.zero_()
        code = np.where(lab < 0, 255, lab - label_start + 1).astype(np.uint8)
        out.numpy()[pts["vox"].numpy()[:n]] = code

    def meta_bytes(self):
        import ctypes
        from stemseg_amd import hip
        return ctypes.sizeof(hip.ClusterMeta)

    def pack_meta(self, m):
        import ctypes
        from stemseg_amd import hip
        rec = hip.ClusterMeta()
        rec.K = m.K
        for i in range(m.K):
            for e in range(8):
                rec.centers[i][e] = m.centers[i][e]
                rec.bandwidths[i][e] = m.bandwidths[i][e]
        return torch.from_numpy(np.frombuffer(bytes(rec), np.uint8).copy())

    def unpack_meta(self, raw):
        from stemseg_amd import hip
        return hip.ClusterMeta.from_buffer_copy(bytes(raw))

    def fg_from_semseg(self, per_clip, n_frames, resize_scale):
        """inference_model.py:121-128, 197-231 + inference/main.py:142-144 in torch CPU ops."""
        F_ = torch.nn.functional
        acc, counts = None, [0] * n_frames
        for frames, plane in per_clip:
            x = plane.float()
            if float(resize_scale) != 1.0:
                x = F_.interpolate(x[None], scale_factor=(1.0, resize_scale, resize_scale), mode="trilinear", align_corners=False)[0]
            if acc is None:
                acc = [torch.zeros((x.shape[0],) + tuple(x.shape[2:])) for _ in range(n_frames)]
            for i, t in enumerate(frames):
                acc[t] = acc[t] + x[:, i]
                counts[t] += 1
        logits = torch.stack([a / float(c) for a, c in zip(acc, counts)], 0)          # [F, Cfg, H, W]
        prob = logits[:, 0].sigmoid() if logits.shape[1] == 1 else F_.softmax(logits, dim=1)[:, 1]
        return (prob > 0.5).to(torch.uint8)

    def overflow_byte(self, block):
        ts = block if isinstance(block, (tuple, list)) else [block]
        return torch.tensor(0 if all(bool(torch.isfinite(t).all()) for t in ts) else 1, dtype=torch.uint8)

    @staticmethod
    def _bins(c, B):
        c = c.astype(np.int64)
        return np.where(c == 255, B - 1, np.minimum(c, B - 1))

    def pair_tables(self, codes, plane_a, plane_b, B):
        c = codes.numpy()
        out = np.zeros((len(plane_b), B, B), np.int32)
        for k, (pa, pb) in enumerate(zip(plane_a, plane_b)):
            b = c[pb]
            m = b != 0
            a = c[pa][m] if pa >= 0 else np.zeros(int(m.sum()), np.uint8)
            out[k] = np.bincount(self._bins(a, B) * B + self._bins(b[m], B), minlength=B * B).reshape(B, B)
        return torch.from_numpy(out)

    def read_back(self, *tensors):
        return [t.numpy().copy() for t in tensors]

    def labels_from_codes(self, codes, vox, items, lut, max_count, n_out):
        c, v = codes.numpy(), vox.numpy().astype(np.int64)
        B = lut.shape[1]
        out = np.zeros(n_out, np.int64)
        for (src, cnt, vbase, plane, dst), l in zip(np.asarray(items).tolist(), np.asarray(lut)):
            out[dst:dst + cnt] = l[self._bins(c[plane][v[src:src + cnt] - vbase], B)]
        return torch.from_numpy(out)
