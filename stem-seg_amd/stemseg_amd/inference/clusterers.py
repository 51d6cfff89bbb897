"""Sequential seed-and-threshold clustering on the HIP kernels.

Drop-in for ``stemseg.inference.clusterers`` (ClustererBase :7-31, SequentialClustering :34-175): same
constructor (inference/main.py:84-91), same call
``clusterer(embeddings[N,E], bandwidths=[N,Ev], seediness=[N,1], cluster_label_start=int, return_label_masks=bool)
-> (LongTensor[N] on the input's device, dict)`` (online_chainer.py:283-286), same ``reset_time_log`` /
``average_time``.  The <= 20 rounds of the reference's Python loop become max_instances + 2 kernel launches with a
single host read-back (K and the instance list) at the end.
"""
import time
from collections import defaultdict

import torch

from .. import hip


class ClustererBase(object):
    """Callable with a per-input-size wall-clock log -- the surface ``inference/main.py`` and the chainer use
    (clusterers.py:7-31): ``clusterer(embeddings, ...)``, ``reset_time_log()``, ``average_time``, ``name``."""

    _name = "clusterer"

    def __init__(self):
        self.reset_time_log()

    def reset_time_log(self):
        self._time_log = defaultdict(list)              # number of points -> [seconds per call]

    def __call__(self, embeddings, *args, **kwargs):
        if embeddings.dtype != torch.float32:
            raise AssertionError("embeddings must be float32, got %s" % embeddings.dtype)
        started = time.time()
        result = self._process(embeddings, *args, **kwargs)
        self._time_log[int(embeddings.shape[0])].append(time.time() - started)
        return result

    def _process(self, embeddings, *args, **kwargs):
        raise NotImplementedError("Must be implemented by derived class")

    @property
    def average_time(self):
        calls = [t for ts in self._time_log.values() for t in ts]
        return sum(calls) / float(len(calls))

    @property
    def name(self):
        return self._name


class SequentialClustering(ClustererBase):
    _name = "SequentialClustering"

    def __init__(self, primary_prob_thresh, secondary_prob_thresh, min_seediness_prob, n_free_dims, free_dim_stds, device,
                 max_instances=20):
        """Same arguments as the reference (clusterers.py:35-50; built at inference/main.py:84-91).  ``device`` is where inputs
        that arrive on the host are moved for clustering."""
        super().__init__()
        if n_free_dims and len(free_dim_stds) < n_free_dims:
            raise AssertionError("%d free dims but only %d stds" % (n_free_dims, len(free_dim_stds)))
        self.thresholding_mode = "probability"
        self.primary_prob_thresh, self.secondary_prob_thresh = primary_prob_thresh, secondary_prob_thresh
        self.min_seediness_prob, self.max_instances = min_seediness_prob, max_instances
        self.n_free_dims, self.free_dim_stds = n_free_dims, list(free_dim_stds)
        self.device = device

    def _params(self):
        return hip.make_cluster_params(self.primary_prob_thresh, self.secondary_prob_thresh, self.min_seediness_prob,
                                       self.max_instances, self.free_dim_stds[:self.n_free_dims] if self.n_free_dims else [])

    # ---- asynchronous form used by the clip pipeline: no host sync, N may live on the device ---------------
    @torch.no_grad()
    def enqueue(self, embeddings, bandwidths, seediness, cluster_label_start=1, n_points_dev=None,
                return_label_masks=False, return_probs=False):
        """Device tensors in; returns (labels[Nmax] int64, meta_dev, masks, probs) without synchronising."""
        hip.require_gpu()
        return hip.cluster(embeddings.contiguous(), bandwidths.contiguous(), seediness.reshape(-1).contiguous(), self._params(),
                           cluster_label_start, n_points_dev, return_label_masks, return_probs)

    @torch.no_grad()
    def enqueue_batch(self, point_sets, cluster_label_start=1):
        """The clips of one step at once: point_sets = [(embeddings [Nmax,E], bandwidths [Nmax,Ev], seediness [Nmax], n_points_dev)];
        one sequence of max_instances + 2 launches serves all of them (stemseg_hip_cluster_batch).  -> [(labels, meta_dev)]."""
        hip.require_gpu()
        return hip.cluster_batch([(e.contiguous(), b.contiguous(), s.reshape(-1).contiguous(), n) for e, b, s, n in point_sets],
                                 self._params(), cluster_label_start)

    def meta_to_dict(self, meta, E, label_start, masks=None, probs=None, n=None):
        K = meta.K
        out = {"instance_labels": [label_start + i for i in range(K)],
               "instance_centers": [[float(meta.centers[i][e]) for e in range(E)] for i in range(K)],
               "instance_stds": [], "instance_masks": []}
        for i in range(K):
            bw = torch.tensor([meta.bandwidths[i][e] for e in range(E)], dtype=torch.float32)
            out["instance_stds"].append((1. / bw).clamp(min=1e-8).sqrt().tolist())       # clusterers.py:124
        if masks is not None:
            out["instance_masks"] = [masks[i, :n].bool().cpu() for i in range(K)]         # primary masks (A.2 quirk iii)
        if probs is not None:
            out["instance_probs"] = [probs[i, :n].cpu() for i in range(K)]
        return out

    @torch.no_grad()
    def _process(self, embeddings, bandwidths, seediness, cluster_label_start=1, *args, **kwargs):
        if embeddings.numel() == 0:
            return torch.zeros(0, dtype=torch.long, device=embeddings.device), \
                {'instance_labels': [], 'instance_centers': [], 'instance_stds': [], 'instance_masks': []}
        input_device = embeddings.device
        embeddings = embeddings.to(device=self.device)
        assert torch.is_tensor(bandwidths) and torch.is_tensor(seediness)
        if bandwidths.shape[0] != embeddings.shape[0]:
            bandwidths = bandwidths.expand_as(embeddings)
        bandwidths = bandwidths.to(device=self.device)
        if self.n_free_dims == 0:
            assert embeddings.shape == bandwidths.shape
        seediness = seediness.to(device=self.device)
        want_masks = kwargs.get("return_label_masks", False)
        want_probs = kwargs.get("return_probs", False)
        labels, meta_dev, masks, probs = self.enqueue(embeddings, bandwidths, seediness, cluster_label_start, None,
                                                       want_masks, want_probs)
        meta = hip.read_cluster_meta(meta_dev)          # the one host read-back
        n = embeddings.shape[0]
        info = self.meta_to_dict(meta, embeddings.shape[1], cluster_label_start, masks, probs, n)
        return labels.to(input_device), info
