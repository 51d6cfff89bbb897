from stemseg_amd.modeling.inference_model import InferenceModel  # noqa: F401
