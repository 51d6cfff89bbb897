from stemseg_amd.config import cfg, load_preset, make_cfg  # noqa: F401
