"""Where datasets and meta files live, from the same environment variables as the reference (``stemseg/data/paths.py:4-77``:
STEMSEG_JSON_ANNOTATIONS_DIR, DAVIS_BASE_DIR, YOUTUBE_VIS_BASE_DIR, KITTIMOTS_BASE_DIR) and ``utils/constants.py:49-59``
(RepoPaths).  Static classes, never instantiated."""
import os


def _env(name):
    v = os.getenv(name)
    if v is None:
        raise EnvironmentError("Required environment variable '{}' is not set.".format(name))
    return v


class _Static(object):
    def __init__(self):
        raise ValueError("Static class '{}' should not be instantiated".format(type(self).__name__))


class RepoPaths(_Static):
    @staticmethod
    def dataset_meta_info_dir():
        return os.path.realpath(os.path.join(os.path.dirname(__file__), os.pardir, "data", "metainfo"))

    @staticmethod
    def configs_dir():
        return os.path.realpath(os.path.join(os.path.dirname(__file__), os.pardir, os.pardir, "config"))


class DavisUnsupervisedPaths(_Static):
    trainval_base_dir = staticmethod(lambda: _env("DAVIS_BASE_DIR"))
    train_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "davis_train.json"))
    val_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "davis_val.json"))


class YoutubeVISPaths(_Static):
    training_base_dir = staticmethod(lambda: os.path.join(_env("YOUTUBE_VIS_BASE_DIR"), "train"))
    val_base_dir = staticmethod(lambda: os.path.join(_env("YOUTUBE_VIS_BASE_DIR"), "valid"))
    train_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "youtube_vis_train.json"))
    val_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "youtube_vis_val.json"))


class KITTIMOTSPaths(_Static):
    train_images_dir = staticmethod(lambda: _env("KITTIMOTS_BASE_DIR"))
    train_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "kittimots_train.json"))
    val_vds_file = staticmethod(lambda: os.path.join(_env("STEMSEG_JSON_ANNOTATIONS_DIR"), "kittimots_val.json"))
