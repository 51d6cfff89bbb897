from stemseg_amd.utils.video_dataset import GenericVideoSequence, parse_generic_video_dataset  # noqa: F401
