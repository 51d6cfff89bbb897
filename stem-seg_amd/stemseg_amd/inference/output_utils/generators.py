"""Per-dataset output generators with the reference's call contract (``inference/main.py:166-169,248-262``):
``process_sequence(sequence, track_mask_idxes, track_mask_labels, instance_pt_counts, instance_lifetimes, category_masks,
mask_dims, mask_scale, max_tracks, device)`` and ``save()``.

The device-side half (instances to keep, labels -> full-resolution masks) is ``MaskMaterializer``.  The DAVIS writer
(indexed PNG per frame, ``output_utils/davis.py:108-121``) is complete; the YouTube-VIS json (COCO-RLE, per-instance category
voting) and KITTI-MOTS txt formats need pycocotools and are out of scope -- those two classes materialise
the masks and keep them for the caller.
"""
import os

import numpy as np

from .masks import MaskMaterializer


def pascal_color_map(n=256):
    """The PASCAL-VOC / DAVIS palette: bit-interleaved colours, uint8 [n, 3]."""
    cmap = np.zeros((n, 3), np.uint8)
    for i in range(n):
        c, rgb = i, [0, 0, 0]
        for j in range(8):
            for ch in range(3):
                rgb[ch] |= ((c >> ch) & 1) << (7 - j)
            c >>= 3
        cmap[i] = rgb
    return cmap


class _OutputGeneratorBase(object):
    def __init__(self, output_dir, outlier_label, save_visualization, *args, **kwargs):
        self.results_output_dir = os.path.join(output_dir, "results")
        self.outlier_label = outlier_label
        self.save_visualization = save_visualization
        self.upscaled_inputs = bool(kwargs.get("upscaled_inputs"))
        self.sequences = {}

    def _materialize(self, sequence, track_mask_idxes, track_mask_labels, instance_lifetimes, mask_dims, mask_scale, max_tracks, device):
        m = MaskMaterializer(self.outlier_label, self.upscaled_inputs)
        dev = "cuda" if str(device) == "cpu" else device             # the kernels run on the GPU whatever the writer asked for
        return m.process_sequence(sequence.image_dims, track_mask_idxes, track_mask_labels, instance_lifetimes, mask_dims,
                                  mask_scale, max_tracks, dev)

    def save(self, *args, **kwargs):
        pass


class DavisOutputGenerator(_OutputGeneratorBase):
    def process_sequence(self, sequence, track_mask_idxes, track_mask_labels, instance_pt_counts, instance_lifetimes,
                         category_masks, mask_dims, mask_scale, max_tracks, device="cpu"):
        from PIL import Image
        keep, masks = self._materialize(sequence, track_mask_idxes, track_mask_labels, instance_lifetimes, mask_dims, mask_scale,
                                        max_tracks, device)
        out_dir = os.path.join(self.results_output_dir, str(sequence.id))
        os.makedirs(out_dir, exist_ok=True)
        palette = pascal_color_map().flatten().tolist()
        for t, m in enumerate(masks.cpu().numpy()):
            im = Image.fromarray(m)
            im.putpalette(palette)
            im.save(os.path.join(out_dir, "{:05d}.png".format(t)))
        return keep, dict()


class _MasksOnlyGenerator(_OutputGeneratorBase):
    FORMAT = ""

    def process_sequence(self, sequence, track_mask_idxes, track_mask_labels, instance_pt_counts, instance_lifetimes,
                         category_masks, mask_dims, mask_scale, max_tracks, device="cpu"):
        keep, masks = self._materialize(sequence, track_mask_idxes, track_mask_labels, instance_lifetimes, mask_dims, mask_scale,
                                        max_tracks, device)
        self.sequences[sequence.id] = dict(instances=keep, masks=masks, category_masks=category_masks)   # plane n: masks == n + 1
        return keep, dict()

    def save(self, *args, **kwargs):
        raise NotImplementedError("%s serialisation is outside the hot path (SURVEY.md section 2); the materialised masks are in "
                                  "`.sequences[seq_id]`" % self.FORMAT)


class YoutubeVISOutputGenerator(_MasksOnlyGenerator):
    FORMAT = "YouTube-VIS json (COCO-RLE + category voting, output_utils/youtube_vis.py)"

    def __init__(self, output_dir, outlier_label, save_visualization, category_mapping=None, category_names=None, *args, **kwargs):
        super().__init__(output_dir, outlier_label, save_visualization, *args, **kwargs)
        self.category_mapping, self.category_names = category_mapping, category_names


class KittiMOTSOutputGenerator(_MasksOnlyGenerator):
    FORMAT = "KITTI-MOTS txt (RLE, output_utils/kitti_mots.py)"
