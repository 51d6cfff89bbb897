from stemseg_amd.utils.global_registry import GlobalRegistry  # noqa: F401
