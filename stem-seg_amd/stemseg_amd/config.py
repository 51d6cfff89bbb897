"""The handful of configuration keys the hot path reads ('Config / flags'),
with the dataset presets of the reference (config/defaults.yaml, davis_1.yaml, davis_2.yaml, youtube_vis.yaml,
kitti_mots_2.yaml).  ``cfg`` is a process-global like the reference's, but plain and mutable:
decoder topology is fixed from ``cfg.INPUT.NUM_FRAMES`` at module construction (modeling/common.py:15-24).
"""
import copy
from types import SimpleNamespace as NS


def _defaults():
    return NS(
        INPUT=NS(IMAGE_MEAN=[102.9801, 115.9465, 122.7717], IMAGE_STD=[1.0, 1.0, 1.0], MIN_DIM=800, MAX_DIM=1333,
                 NUM_FRAMES=8, NUM_CLASSES=2, BGR_INPUT=True, NORMALIZE_TO_UNIT_SCALE=False),
        MODEL=NS(
            USE_SEMSEG_HEAD=True, USE_SEEDINESS_HEAD=False, EMBEDDING_DIM_MODE="xyt",
            BACKBONE=NS(TYPE="R-101-FPN"),
            RESNETS=NS(BACKBONE_OUT_CHANNELS=256),
            EMBEDDINGS=NS(HEAD_TYPE="squeeze_expand_decoder", INTER_CHANNELS=[256, 256, 128, 128], SCALE=[32, 16, 8, 4],
                          EMBEDDING_SIZE=3, TANH_ACTIVATION=True, NORMALIZATION_LAYER="gn", GN_NUM_GROUPS=32, POOL_TYPE="avg"),
            SEMSEG=NS(HEAD_TYPE="squeeze_expand_decoder", FEATURE_SCALE=[4, 8, 16, 32], INTER_CHANNELS=[256, 256, 128, 128],
                      NORMALIZATION_LAYER="gn", GN_NUM_GROUPS=32, POOL_TYPE="avg", FOREGROUND_CHANNEL=True),
            SEEDINESS=NS(HEAD_TYPE="squeeze_expand_decoder", INTER_CHANNELS=[256, 256, 128, 128], FEATURE_SCALE=[32, 16, 8, 4],
                         NORMALIZATION_LAYER="gn", GN_NUM_GROUPS=32, POOL_TYPE="avg"),
        ),
        TRAINING=NS(LOSS_AT_FULL_RES=False, LOSSES=NS(EMBEDDING=NS(FREE_DIM_STDS=[]))),
        DATA=NS(DAVIS=NS(INFERENCE_FRAME_OVERLAP=6), YOUTUBE_VIS=NS(INFERENCE_FRAME_OVERLAP=4), KITTI_MOTS=NS(INFERENCE_FRAME_OVERLAP=4)),
        CLUSTERING=NS(MIN_SEEDINESS_PROB=0.8, PRIMARY_PROB_THRESHOLD=0.5, SECONDARY_PROB_THRESHOLD=0.3),
    )


def _apply(c, preset):
    if preset in ("davis", "davis_2"):   # config/davis_1.yaml (NUM_FRAMES 8) / davis_2.yaml (NUM_FRAMES 16: what inference/main.py:188-195 loads
        # for --dataset davis when the checkpoint directory holds no config.yaml); the hot-path keys differ in NUM_FRAMES only
        c.INPUT.MIN_DIM, c.INPUT.MAX_DIM, c.INPUT.NUM_FRAMES = 736, 1248, (16 if preset == "davis_2" else 8)
        c.MODEL.EMBEDDING_DIM_MODE, c.MODEL.USE_SEEDINESS_HEAD, c.MODEL.USE_SEMSEG_HEAD = "xyff", True, False
        c.MODEL.EMBEDDINGS.EMBEDDING_SIZE = 4
        c.TRAINING.LOSSES.EMBEDDING.FREE_DIM_STDS = [0.3, 0.3]
    elif preset == "ytvis":         # config/youtube_vis.yaml
        c.INPUT.MIN_DIM, c.INPUT.MAX_DIM, c.INPUT.NUM_FRAMES, c.INPUT.NUM_CLASSES = 640, 1196, 8, 41
        c.MODEL.EMBEDDING_DIM_MODE, c.MODEL.USE_SEEDINESS_HEAD, c.MODEL.USE_SEMSEG_HEAD = "xyff", False, True
        c.MODEL.EMBEDDINGS.EMBEDDING_SIZE = 4
        c.MODEL.SEMSEG.INTER_CHANNELS = [256, 256, 256, 256]
        c.TRAINING.LOSSES.EMBEDDING.FREE_DIM_STDS = [0.3, 0.3]
    elif preset == "kittimots":     # config/kitti_mots_2.yaml
        c.INPUT.MIN_DIM, c.INPUT.MAX_DIM, c.INPUT.NUM_FRAMES, c.INPUT.NUM_CLASSES = 736, 1792, 8, 3
        c.MODEL.EMBEDDING_DIM_MODE, c.MODEL.USE_SEEDINESS_HEAD, c.MODEL.USE_SEMSEG_HEAD = "xyt", False, True
        c.CLUSTERING.MIN_SEEDINESS_PROB = 0.95
    elif preset not in (None, "defaults"):
        raise ValueError("unknown preset '%s' (davis | davis_2 | ytvis | kittimots | defaults)" % preset)
    return c


def make_cfg(preset=None):
    return _apply(_defaults(), preset)


cfg = make_cfg()


def load_preset(preset):
    """Re-initialise the global cfg in place (call BEFORE constructing any module)."""
    new = make_cfg(preset)
    for k, v in vars(new).items():
        setattr(cfg, k, copy.deepcopy(v))
    return cfg


# ---- external configuration sources (stemseg_amd.overlay: the reference's own global cfg) -----------------------------
_sources = []


def register_source(fn):
    """fn(cfg) copies values into ``cfg``; called by ``refresh()`` -- i.e. whenever a model is built or run."""
    if fn not in _sources:
        _sources.append(fn)


def unregister_source(fn):
    if fn in _sources:
        _sources.remove(fn)


def refresh():
    for fn in _sources:
        fn(cfg)
    return cfg
