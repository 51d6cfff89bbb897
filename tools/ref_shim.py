"""Import shim for the *reference* (read-only, /root/reference).  Runs ONLY in the build container:
nothing under tests/, bench.py, __graft_entry__.py or the product package imports this file, and the
reference itself never travels to the GPU box.

What it does (and why):
  1. yaml.load defaults to FullLoader  -- reference bug on PyYAML >= 6 (config/config.py:186-194)
  2. stubs cv2 / imgaug / pycocotools / tensorboardX in sys.modules -- imported transitively by
     stemseg/data/__init__.py and inference/output_utils/__init__.py, never *called* on this path
  3. makes .cuda() the identity -- hard-coded device moves (inference_model.py:102,
     online_chainer.py:174-176,299-302, inference/main.py:67,97); there is no GPU here
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Permissive(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return self

    def __mro_entries__(self, bases):       # allow `class X(stub.Something)`
        return (object,)


def install(num_frames=None):
    assert os.path.isdir(REFERENCE_ROOT), "reference tree not present (this tool only runs in the build container)"
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True

    import yaml
    if not getattr(yaml, "_stemseg_shimmed", False):
        _orig = yaml.load

        def _load(stream, Loader=None):
            return _orig(stream, Loader=Loader or yaml.FullLoader)
        yaml.load = _load
        yaml._stemseg_shimmed = True

    for m in ("cv2", "imgaug", "imgaug.augmenters", "imgaug.augmentables", "imgaug.augmentables.segmaps",
              "pycocotools", "pycocotools.mask", "tensorboardX"):
        if m not in sys.modules:
            sys.modules[m] = _Permissive(m)

    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    from stemseg.config import cfg
    if num_frames is not None:
        cfg.INPUT.update_param("NUM_FRAMES", int(num_frames))
    from stemseg.structures import ImageList
    ImageList.cuda = lambda self, *a, **k: self
    return cfg
