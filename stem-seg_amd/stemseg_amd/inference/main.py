"""Clip windowing, fg mask from seediness, and the per-sequence driver.

Counterpart of ``stemseg/inference/main.py``: get_subsequence_frames :23-49, TrackGenerator :52-170
(create_clusterer :84-91, get_fg_masks_from_seediness :93-103, do_inference :132-149, do_clustering :151-170).
Dataset parsing, CLI and the per-dataset output writers of the reference are out of scope.
"""
import torch

from .. import hip
from ..config import cfg
from ..modeling.embedding_utils import get_nb_free_dims
from .clusterers import SequentialClustering
from .online_chainer import OnlineChainer

_OVERLAP_KEY = {"davis": "DAVIS", "ytvis": "YOUTUBE_VIS", "kittimots": "KITTI_MOTS"}


def get_subsequence_frames(seq_len, subseq_len, dataset_name, frame_overlap=-1):
    """-> (list of frame-index lists, padded-flags or None).  Stride subseq_len - overlap, plus one tail clip that
    ends on the last frame; videos shorter than a clip repeat frame 0 on the left."""
    if dataset_name not in _OVERLAP_KEY:
        raise NotImplementedError()
    if frame_overlap <= 0:                     # 0 and negatives mean "dataset default" (main.py:27-31)
        frame_overlap = getattr(cfg.DATA, _OVERLAP_KEY[dataset_name]).INFERENCE_FRAME_OVERLAP
    assert frame_overlap < subseq_len
    if seq_len < subseq_len:
        n_pad = subseq_len - seq_len
        return [[0] * n_pad + list(range(seq_len))], [True] * n_pad + [False] * seq_len
    starts = list(range(0, seq_len - subseq_len + 1, subseq_len - frame_overlap))
    clips = [list(range(t, t + subseq_len)) for t in starts]
    if not clips or clips[-1][-1] != seq_len - 1:
        clips.append(list(range(seq_len - subseq_len, seq_len)))
    return clips, None


@torch.no_grad()
def fg_masks_from_seediness(embedding_maps, threshold, device=None):
    """Mean seediness over the clips containing each frame, > threshold -> uint8 [n_frames, h, w] on the device
    (inference/main.py:93-103).  One accumulate launch per clip (its T planes go to their frames' sums; clip order = the
    reference's ``+=`` order) and ONE mask launch for the whole sequence."""
    hip.require_gpu()
    frames_all = sorted({t for entry in embedding_maps for t in entry[0]})
    index = {t: i for i, t in enumerate(frames_all)}
    acc, counts = None, [0.0] * len(frames_all)
    for entry in embedding_maps:
        frames, seed = list(entry[0]), entry[3]
        seed = (seed if seed.is_cuda else seed.to(device if device is not None else "cuda")).contiguous().float()
        if acc is None:
            acc = torch.zeros((len(frames_all), 1) + tuple(seed.shape[-2:]), dtype=torch.float32, device=seed.device)
        assert len(set(frames)) == len(frames) and seed.shape[1] == len(frames)
        hip.semseg_accumulate(acc, seed, [index[t] for t in frames])
        for t in frames:
            counts[index[t]] += 1.0
    if acc is None:
        return torch.zeros(0, dtype=torch.uint8)
    return hip.fg_mask_frames(acc[:, 0], torch.tensor(counts, dtype=torch.float32).to(acc.device), threshold)


class TrackGenerator(object):
    """Sequence -> clips -> head outputs -> fg mask -> clustering + stitching (no dataset / writer plumbing)."""

    def __init__(self, model, dataset_name, resize_scale=1.0, **kwargs):
        self.model = model
        self.dataset_name = dataset_name
        self.resize_scale = resize_scale
        self.seediness_fg_threshold = kwargs.get("seediness_thresh", 0.25)
        self.frame_overlap = kwargs.get("frame_overlap", -1)
        self.clustering_device = kwargs.get("clustering_device", "cuda:0")
        ops = kwargs.get("ops")
        if ops is None:
            from .online_chainer import HipChainerOps
            ops = HipChainerOps(self.clustering_device)
        self.chainer = OnlineChainer(self.create_clusterer(), embedding_resize_factor=resize_scale, ops=ops)

    def create_clusterer(self):
        c = cfg.CLUSTERING
        return SequentialClustering(primary_prob_thresh=c.PRIMARY_PROB_THRESHOLD, secondary_prob_thresh=c.SECONDARY_PROB_THRESHOLD,
                                    min_seediness_prob=c.MIN_SEEDINESS_PROB, n_free_dims=get_nb_free_dims(cfg.MODEL.EMBEDDING_DIM_MODE),
                                    free_dim_stds=cfg.TRAINING.LOSSES.EMBEDDING.FREE_DIM_STDS, device=self.clustering_device)

    def get_fg_masks_from_seediness(self, inference_output):
        return fg_masks_from_seediness(inference_output['embeddings'], self.seediness_fg_threshold, self.clustering_device)

    def do_inference(self, frames):
        n = len(frames)
        subseq_idxes, _ = get_subsequence_frames(n, cfg.INPUT.NUM_FRAMES, self.dataset_name, self.frame_overlap)
        out = self.model(frames, subseq_idxes)
        fg_masks = out["fg_masks"]
        if torch.is_tensor(fg_masks):            # semseg head present: its foreground probability > 0.5 (main.py:142-144)
            fg_masks = fg_masks if fg_masks.is_cuda else fg_masks.to(self.clustering_device)
            fg_masks = torch.stack([hip.fg_mask(p.contiguous(), 1.0, 0.5) for p in fg_masks], 0)
        else:                                    # otherwise the seediness map, averaged over clips, > threshold (:145-147)
            fg_masks = self.get_fg_masks_from_seediness(out)
        return out["embeddings"], fg_masks, out["multiclass_masks"]

    def do_clustering(self, all_embeddings, fg_masks):
        dicts = [{"frames": f, "embeddings": e, "bandwidths": b, "seediness": s} for (f, e, b, s) in all_embeddings]
        return self.chainer.process(fg_masks, dicts)

    def process_sequence(self, frames):
        embeddings, fg_masks, _ = self.do_inference(frames)
        return self.do_clustering(embeddings, fg_masks)
