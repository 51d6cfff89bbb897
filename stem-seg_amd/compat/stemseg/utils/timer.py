from stemseg_amd.utils.timer import Timer  # noqa: F401
