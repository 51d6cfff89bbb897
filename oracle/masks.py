"""Oracle: mask materialisation right after the stitched labels, plain torch CPU ops.

Follows /root/reference/stemseg/inference/output_utils/davis.py:38-116 (the YouTube-VIS and KITTI-MOTS writers run the same
resampling chain, youtube_vis.py:118-155, kitti_mots.py:89-130) and data/common.py:142-159.  TEST INFRASTRUCTURE -- see
oracle/__init__.py.  File formats (PNG palette, RLE, MOTS txt) are not restated.
"""
import numpy as np
import torch
import torch.nn.functional as F


def compute_resize_params_2(image_dims, min_dim, max_dim):
    """data/common.py:142-159: (width, height) -> (new_width, new_height, scale)."""
    lower, higher = float(min(image_dims)), float(max(image_dims))
    scale = min_dim / lower
    if higher * scale > max_dim:
        scale = max_dim / higher
    return round(scale * image_dims[0]), round(scale * image_dims[1]), scale


def instances_to_keep(lifetimes, outlier_label, max_tracks):
    """davis.py:57-66: ids by descending lifetime (stable: ties keep the dict's insertion order), outliers dropped, first
    max_tracks."""
    ranked = sorted([(k, v) for k, v in lifetimes.items()], key=lambda x: x[1], reverse=True)
    return [k for k, _ in ranked if k != outlier_label][:max_tracks]


@torch.no_grad()
def condensed_masks(label_maps, keep, image_hw, min_dim, max_dim, mask_scale=4.0, upscaled_inputs=False):
    """label_maps int64 [F, h, w] (0 = background; davis.py:76-77 scatters the per-point labels into zeros) ->
    uint8 [F, image_h, image_w]: value n + 1 where instance keep[n] covers the pixel, else 0 (davis.py:76-116)."""
    maps = torch.as_tensor(np.asarray(label_maps))
    ih, iw = image_hw
    out = []
    for t in range(maps.shape[0]):
        m = torch.stack([maps[t] == i for i in keep], 0).unsqueeze(0).float() if len(keep) else torch.zeros(1, 0, *maps.shape[1:])
        if not upscaled_inputs:
            m = F.interpolate(m, scale_factor=mask_scale, mode="bilinear", align_corners=False)
        rw, rh, _ = compute_resize_params_2((iw, ih), min_dim, max_dim)
        assert m.shape[3] >= rw and m.shape[2] >= rh
        m = m[:, :, :rh, :rw]
        m = (F.interpolate(m, (ih, iw), mode="bilinear", align_corners=False) > 0.5)[0]
        cond = torch.zeros(ih, iw, dtype=torch.uint8)
        for n in range(len(keep)):
            cond = torch.where(m[n], torch.tensor(n + 1, dtype=torch.uint8), cond)
        out.append(cond)
    return torch.stack(out, 0)


@torch.no_grad()
def soft_masks(label_map, keep, image_hw, min_dim, max_dim, mask_scale=4.0):
    """The float planes just before the > 0.5 threshold, [K, image_h, image_w] (used by tests to find pixels that sit on the
    threshold)."""
    m = torch.stack([torch.as_tensor(np.asarray(label_map)) == i for i in keep], 0).unsqueeze(0).float()
    m = F.interpolate(m, scale_factor=mask_scale, mode="bilinear", align_corners=False)
    rw, rh, _ = compute_resize_params_2((image_hw[1], image_hw[0]), min_dim, max_dim)
    return F.interpolate(m[:, :, :rh, :rw], image_hw, mode="bilinear", align_corners=False)[0]
