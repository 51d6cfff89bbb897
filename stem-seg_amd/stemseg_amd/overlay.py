"""Import overlay: run code written against sabarim/STEm-Seg (``import stemseg...``) with the MI355X hot path underneath.

The reference has no FFI or plug-in ABI: its seam is Python modules.  ``install()`` puts one finder
at the head of ``sys.meta_path`` that answers ONLY for the hot-path modules below -- each resolves to the ``stemseg_amd``
module with the same public names -- and lets every other ``stemseg.*`` import fall through to whatever ``stemseg`` package
is on ``sys.path`` (a reference checkout: its ``inference/main.py``, ``data``, ``config``, ``utils``, ``output_utils`` run
unchanged).  Without a checkout, ``stem-seg_amd/compat`` (appended to ``sys.path`` as the LAST resort) supplies a minimal
``stemseg`` skeleton for the non-hot-path names that ``inference/main.py:5-18`` imports.

    PYTHONPATH=/path/to/STEm-Seg python -m stemseg_amd.overlay /path/to/STEm-Seg/stemseg/inference/main.py <its args>
    # or, inside a program:   import stemseg_amd.overlay as ov; ov.install(); from stemseg.inference.main import ...

The reference's global ``stemseg.config.cfg`` stays the single source of configuration: ``sync_cfg()`` copies the keys the
hot path reads into ``stemseg_amd.config.cfg`` whenever a model is built or run (``config.refresh``).
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

# reference module (file under /stemseg)                      -> drop-in here
HOT_PATH = {
    "stemseg.modeling.embedding_decoder": "stemseg_amd.modeling.embedding_decoder",     # modeling/embedding_decoder.py
    "stemseg.modeling.seediness_decoder": "stemseg_amd.modeling.seediness_decoder",     # modeling/seediness_decoder.py
    "stemseg.modeling.semseg_decoder": "stemseg_amd.modeling.semseg_decoder",           # modeling/semseg_decoder.py
    "stemseg.modeling.embedding_utils": "stemseg_amd.modeling.embedding_utils",         # modeling/embedding_utils.py
    "stemseg.modeling.common": "stemseg_amd.modeling.common",                           # modeling/common.py
    "stemseg.modeling.model_builder": "stemseg_amd.modeling.model_builder",             # modeling/model_builder.py (build_model)
    "stemseg.modeling.inference_model": "stemseg_amd.modeling.inference_model",         # modeling/inference_model.py
    "stemseg.inference.clusterers": "stemseg_amd.inference.clusterers",                 # inference/clusterers.py
    "stemseg.inference.online_chainer": "stemseg_amd.inference.online_chainer",         # inference/online_chainer.py
}
COMPAT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)       # the SAME module object under both names

    def exec_module(self, module):
        pass


class _HotPathFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        t = HOT_PATH.get(fullname)
        if t is None:
            return None                                    # not ours: the next finder (the reference checkout) answers
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(t), origin="stemseg_amd overlay -> " + t)


_finder = None


def install(skeleton=True):
    """Idempotent.  ``skeleton``: append stem-seg_amd/compat to sys.path so that ``stemseg.{config,utils,data,...}`` resolve
    even without a reference checkout (a checkout earlier on sys.path always wins)."""
    global _finder
    if _finder is None:
        _finder = _HotPathFinder()
        sys.meta_path.insert(0, _finder)
        for name in HOT_PATH:                              # a copy imported before install() would bypass the overlay
            if name in sys.modules and not getattr(sys.modules[name], "__name__", "").startswith("stemseg_amd"):
                del sys.modules[name]
        from . import config
        config.register_source(sync_cfg)
    if skeleton and COMPAT_DIR not in sys.path:
        sys.path.append(COMPAT_DIR)
    return _finder


def uninstall():
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
        for name in list(sys.modules):
            if name in HOT_PATH:
                del sys.modules[name]
        from . import config
        config.unregister_source(sync_cfg)
    if COMPAT_DIR in sys.path:
        sys.path.remove(COMPAT_DIR)


def reference_present():
    """True when the ``stemseg`` package on sys.path is a reference checkout (it has the training side) and not the skeleton."""
    spec = importlib.util.find_spec("stemseg")
    if spec is None or not spec.submodule_search_locations:
        return False
    return any(os.path.isdir(os.path.join(p, "training")) for p in spec.submodule_search_locations)


def sync_cfg(dst):
    """Copy every key of ``stemseg_amd.config.cfg`` (the hot path's subset, same names) from the reference's
    ``stemseg.config.cfg`` (config/config.py: a YamlConfig tree, attribute access) -- the reference's cfg is what
    ``inference/main.py:174-233`` loads and edits."""
    ref = sys.modules.get("stemseg.config")
    src = getattr(ref, "cfg", None)
    if src is None or src is dst:
        return

    def walk(d, s):
        for k, v in vars(d).items():
            if not hasattr(s, k):
                continue
            sv = getattr(s, k)
            if hasattr(v, "__dict__") and not isinstance(v, (list, tuple)):
                walk(v, sv)
            else:
                setattr(d, k, list(sv) if isinstance(sv, (list, tuple)) else sv)
    walk(dst, src)


def main(argv=None):
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    if argv[0] == "-m":
        sys.argv = argv[1:]
        runpy.run_module(argv[1], run_name="__main__", alter_sys=True)
    else:
        sys.argv = argv
        runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
