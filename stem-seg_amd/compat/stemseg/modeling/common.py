from stemseg_amd.modeling.common import UpsampleTrilinear3D, get_pooling_layer_creator, get_temporal_scales  # noqa: F401
