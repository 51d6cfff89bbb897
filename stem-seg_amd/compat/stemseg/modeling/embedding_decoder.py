from stemseg_amd.modeling.embedding_decoder import EMBEDDING_HEAD_REGISTRY, SqueezingExpandDecoder  # noqa: F401
