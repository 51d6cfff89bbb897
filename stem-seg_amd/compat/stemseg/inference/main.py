from stemseg_amd.inference.main import TrackGenerator, get_subsequence_frames  # noqa: F401
