from stemseg_amd.modeling.seediness_decoder import SEEDINESS_HEAD_REGISTRY, SqueezingExpandDecoder  # noqa: F401
