"""The inference-side view of the reference's generic video-dataset JSON (``stemseg/data/generic_video_dataset_parser.py:9-59``):
``{"meta": {"category_labels": {...}}, "sequences": [{"id", "height", "width", "image_paths": [...], ...}]}``.  Only what
``inference/main.py`` touches is kept (paths, dims, id, length); annotation decoding (RLE masks) belongs to training."""
import json


class GenericVideoSequence(object):
    def __init__(self, seq_dict, base_dir):
        self.base_dir = base_dir
        self.image_paths = list(seq_dict["image_paths"])
        self.image_dims = (seq_dict["height"], seq_dict["width"])
        self.id = self.seq_id = seq_dict["id"]
        self.instance_categories = {int(k): v for k, v in seq_dict.get("categories", {}).items()} or None

    def __len__(self):
        return len(self.image_paths)


def parse_generic_video_dataset(base_dir, dataset_json):
    with open(dataset_json, "r") as fh:
        dataset = json.load(fh)
    meta = dataset["meta"]
    meta["category_labels"] = {int(k): v for k, v in meta["category_labels"].items()}
    return [GenericVideoSequence(s, base_dir) for s in dataset["sequences"]], meta
