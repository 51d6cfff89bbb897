"""Labels -> masks in the original image size, on the device.

Mirrors ``DavisOutputGenerator.process_sequence`` (output_utils/davis.py:38-116) up to the point where it hands the
condensed uint8 map to PIL; ``YoutubeVISOutputGenerator`` / ``KittiMOTSOutputGenerator`` run the same chain per instance
(youtube_vis.py:118-155, kitti_mots.py:89-130) -- their binary planes are ``condensed == n + 1``.
Two HIP launches per frame (scatter, fused resample); nothing is synchronised.
"""
import torch

from ... import hip
from ... import config as _config
from ...config import cfg
from ...modeling.inference_model import compute_resize_params_2


def instances_to_keep(instance_lifetimes, outlier_label, max_tracks):
    """Ids by descending lifetime -- stable, so ties keep the dict's order -- without the outlier id, first ``max_tracks``
    (davis.py:57-66)."""
    ranked = sorted([(k, v) for k, v in instance_lifetimes.items()], key=lambda x: x[1], reverse=True)
    return [k for k, _ in ranked if k != outlier_label][:max_tracks]


class MaskMaterializer(object):
    def __init__(self, outlier_label=-1, upscaled_inputs=False):
        self.outlier_label = outlier_label
        self.upscaled_inputs = upscaled_inputs

    def _lut(self, keep, device):
        """(label + 1) -> index in ``keep`` + 1, 0 for everything else (built on the host: a few dozen integers)."""
        top = max([k for k in keep if k >= 0] + [0])
        lut = [0] * (top + 2)
        for n, k in enumerate(keep):
            if k >= 0:
                lut[k + 1] = n + 1
        return torch.tensor(lut, dtype=torch.int32).to(device)

    @torch.no_grad()
    def process_sequence(self, image_dims, track_mask_idxes, track_mask_labels, instance_lifetimes, mask_dims, mask_scale=4.0,
                         max_tracks=10, device="cuda"):
        """image_dims (height, width) of the original frames; track_mask_idxes[t] = (ys, xs) int64 tensors of frame t's
        foreground points at mask resolution ``mask_dims`` (h, w); track_mask_labels[t] their stitched track ids.
        Returns (instances_to_keep, uint8 [F, image_height, image_width] on the device: n + 1 where instance
        instances_to_keep[n] covers the pixel)."""
        hip.require_gpu()
        _config.refresh()
        assert len(track_mask_idxes) == len(track_mask_labels)
        assert max_tracks < 256
        mask_h, mask_w = mask_dims
        image_h, image_w = image_dims
        keep = instances_to_keep(instance_lifetimes, self.outlier_label, max_tracks)
        lut = self._lut(keep, device)
        rw, rh, _ = compute_resize_params_2((image_w, image_h), cfg.INPUT.MIN_DIM, cfg.INPUT.MAX_DIM)
        scale = 1.0 if self.upscaled_inputs else mask_scale
        out = []
        for (ys, xs), labels in zip(track_mask_idxes, track_mask_labels):
            ys, xs, labels = (t.to(device=device, dtype=torch.int64).contiguous() for t in (ys, xs, labels))
            dense = hip.scatter_instance_index(ys, xs, labels, lut, mask_h, mask_w)
            out.append(hip.resample_instance_masks(dense, scale, (rh, rw), (image_h, image_w)))
        return keep, (torch.stack(out, 0) if out else torch.zeros(0, image_h, image_w, dtype=torch.uint8, device=device))
