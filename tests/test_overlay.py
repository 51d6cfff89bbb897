"""The drop-in boundary at the Python level: ``stemseg_amd.overlay`` serves the hot-path modules under
the reference's import paths and lets everything else fall through to a sabarim/STEm-Seg checkout -- or, without one, to the
skeleton in stem-seg_amd/compat.  Each scenario runs in a fresh interpreter (the overlay edits sys.meta_path / sys.modules).

  * with the checkout (build container only, skipped elsewhere): the reference's OWN ``stemseg/inference/main.py`` is imported
    unchanged; its TrackGenerator is constructed with the MI355X InferenceModel / chainer / clusterer inside; the cfg bridge
    reproduces the three presets from the reference's yaml files.
  * without it: the import list of ``inference/main.py:5-18`` and ``modeling/inference_model.py:1-5`` (written out below)
    resolves; on the GPU the flow is driven through those names, image files in, DAVIS PNGs out.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

# inference/main.py:5-18 and modeling/inference_model.py:1-5, name for name (cv2 / tqdm / torch lines left out)
IMPORT_LIST = """
from stemseg.config import cfg
from stemseg.inference.output_utils import YoutubeVISOutputGenerator, DavisOutputGenerator, KittiMOTSOutputGenerator
from stemseg.inference.online_chainer import OnlineChainer
from stemseg.inference.clusterers import SequentialClustering
from stemseg.data.generic_video_dataset_parser import parse_generic_video_dataset
from stemseg.data import DavisUnsupervisedPaths as DavisPaths, YoutubeVISPaths, KITTIMOTSPaths
from stemseg.modeling.inference_model import InferenceModel
from stemseg.modeling.embedding_utils import get_nb_free_dims
from stemseg.utils import Timer, RepoPaths
from stemseg.modeling.model_builder import build_model
from stemseg.utils.timer import Timer as Timer2
"""


def _run(code, extra_path=()):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "stem-seg_amd")] + list(extra_path)),
               PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "overlay scenario failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "stemseg")), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("preset,yaml_name,mode", [("davis", "davis_1.yaml", "xyff"), ("ytvis", "youtube_vis.yaml", "xyff"),
                                                   ("kittimots", "kitti_mots_2.yaml", "xyt")])
def test_reference_main_runs_on_the_overlay_unchanged(tmp_path, preset, yaml_name, mode):
    out = _run("""
        import os, sys
        import ref_shim                                       # cv2 / pycocotools / imgaug stubs, yaml loader fix, .cuda() no-op
        import stemseg_amd.overlay as ov
        ov.install()
        ref_cfg = ref_shim.install()
        import stemseg.inference.main as rmain                # the REFERENCE's file, untouched
        assert rmain.__file__.startswith("%s"), rmain.__file__
        assert ov.reference_present()
        import stemseg_amd.inference.clusterers as c, stemseg_amd.inference.online_chainer as oc
        import stemseg_amd.modeling.inference_model as im, stemseg_amd.modeling.embedding_utils as eu
        assert rmain.SequentialClustering is c.SequentialClustering and rmain.OnlineChainer is oc.OnlineChainer
        assert rmain.InferenceModel is im.InferenceModel and rmain.get_nb_free_dims is eu.get_nb_free_dims
        import stemseg.modeling, stemseg.modeling.semseg_decoder as sd, stemseg_amd.modeling.model_builder as mb
        assert stemseg.modeling.build_model is mb.build_model and sd.__name__ == "stemseg_amd.modeling.semseg_decoder"
        # not overlaid -> the checkout's own modules
        import stemseg.data, stemseg.utils, stemseg.inference.output_utils as ou
        assert all(m.__file__.startswith("%s") for m in (stemseg.data, stemseg.utils, ou))
        # cfg bridge: the reference's yaml presets, merged by the reference's own loader, reproduce this repo's presets
        from stemseg_amd import config
        preset, yaml_name, mode = "%s", "%s", "%s"
        ref_cfg.merge_from_file(os.path.join(rmain.RepoPaths.configs_dir(), yaml_name))
        config.load_preset("defaults")
        config.refresh()
        def flat(ns, pre=""):
            o = {}
            for k, v in vars(ns).items():
                o.update(flat(v, pre + k + ".") if hasattr(v, "__dict__") else {pre + k: v})
            return o
        got, exp = flat(config.cfg), flat(config.make_cfg(preset))
        diff = {k: (got[k], exp[k]) for k in exp if got[k] != exp[k]}
        assert not diff, (preset, diff)
        # the reference's TrackGenerator, constructed as inference/main.py:264-275 does, now holds the MI355X classes
        tg = rmain.TrackGenerator([], preset, None, "%s", None, 10, False, 1.0, True, seediness_thresh=0.25, frame_overlap=-1,
                                  clustering_device="cuda:0")
        assert type(tg.model) is im.InferenceModel and type(tg.chainer) is oc.OnlineChainer
        assert type(tg.chainer.clusterer) is c.SequentialClustering and tg.chainer.clusterer.n_free_dims == eu.get_nb_free_dims(mode)
        assert tg.model.has_semseg_head == (preset != "davis") and tg.model._model.embedding_head.embedding_dim_mode == mode
        print("OVERLAY-REF-OK")
    """ % (REFERENCE, REFERENCE, preset, yaml_name, mode, tmp_path), extra_path=[os.path.join(ROOT, "tools"), REFERENCE])
    assert "OVERLAY-REF-OK" in out


def test_import_list_resolves_on_the_skeleton():
    out = _run("""
        import stemseg_amd.overlay as ov
        ov.install()
        %s
        import stemseg, stemseg_amd
        assert "compat" in stemseg.__file__ and not ov.reference_present()
        import stemseg_amd.inference.clusterers as c, stemseg_amd.modeling.inference_model as im
        assert SequentialClustering is c.SequentialClustering and InferenceModel is im.InferenceModel and Timer is Timer2
        from stemseg.modeling.semseg_decoder import SEMSEG_HEAD_REGISTRY
        from stemseg.modeling.embedding_decoder import EMBEDDING_HEAD_REGISTRY
        from stemseg.modeling.seediness_decoder import SEEDINESS_HEAD_REGISTRY
        from stemseg.modeling.common import UpsampleTrilinear3D, get_temporal_scales, get_pooling_layer_creator
        from stemseg.inference.main import TrackGenerator, get_subsequence_frames
        assert OnlineChainer.OUTLIER_LABEL == -1 and "squeeze_expand_decoder" in SEMSEG_HEAD_REGISTRY
        # Timer decorators behave like the reference's (utils/timer.py): nested log / exclude
        import time
        @Timer.log_duration("inference")
        def outer():
            time.sleep(0.02); inner(); time.sleep(0.02)
        @Timer.exclude_duration("inference")
        def inner():
            time.sleep(0.05)
        outer()
        d = Timer.get_duration("inference")
        assert 0.035 < d < 0.07, d
        ov.uninstall()
        import importlib, sys
        assert "stemseg.inference.clusterers" not in sys.modules
        print("OVERLAY-SKELETON-OK")
    """ % textwrap.indent(IMPORT_LIST, "        ").strip())
    assert "OVERLAY-SKELETON-OK" in out


@pytest.mark.gpu
def test_flow_through_the_reference_names_on_gpu(tmp_path):
    """Image FILES in, DAVIS PNGs out, every class reached through its ``stemseg.*`` name (skeleton mode: the GPU box has no
    checkout); the result equals the direct stemseg_amd path, and the registries / build_model level (what the reference's own
    InferenceModel drives: run_backbone + head(list of [1,C,T,h,w])) reproduces the fused path's head outputs."""
    from PIL import Image
    from tests import synth
    frames = synth.synth_frames(12, 90, 120, seed=5)
    paths = []
    for t, f in enumerate(frames):
        paths.append(str(tmp_path / ("%05d.png" % t)))
        Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(paths[-1])            # files hold RGB; loaders return BGR
    np.save(str(tmp_path / "frames.npy"), frames)
    out = _run("""
        import numpy as np, torch
        import stemseg_amd.overlay as ov
        ov.install()
        %s
        from stemseg.inference.main import TrackGenerator
        from tests import synth
        from stemseg.config import load_preset
        load_preset("davis")
        cfg.INPUT.MIN_DIM, cfg.INPUT.MAX_DIM = 96, 128
        cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
        model = InferenceModel(None, semseg_output_type=None, preload_images=True, resize_scale=1.0, semseg_generation_on_gpu=True)
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 61))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 12.0
        model.load_checkpoint_state(new)
        model = model.cuda()
        paths = [r"%s/%%05d.png" %% t for t in range(12)]
        frames = np.load(r"%s/frames.npy")
        probe = model(paths, [list(range(8))])["embeddings"][0].seediness.flatten()
        thr = float(probe.median())
        cfg.CLUSTERING.MIN_SEEDINESS_PROB = float(probe.quantile(0.75))
        tg = TrackGenerator(model, "davis", seediness_thresh=thr, frame_overlap=4)
        emb_p, fg_p, _ = tg.do_inference(paths)                  # file paths, as inference/main.py:137-138
        emb_a, fg_a, _ = tg.do_inference([f for f in frames])    # arrays
        assert torch.equal(fg_p, fg_a) and all(torch.equal(a.embeddings, b.embeddings) for a, b in zip(emb_p, emb_a))
        (track, counts, life), idx, _, _, meta = tg.do_clustering(emb_p, fg_p)
        class Seq: id, image_dims, image_paths, base_dir = "seq0", (90, 120), paths, ""
        gen = DavisOutputGenerator(r"%s/out", OnlineChainer.OUTLIER_LABEL, False, upscaled_inputs=False)
        keep, _ = gen.process_sequence(Seq, idx, track, counts, life, None, fg_p.shape[-2:], 4.0, 10, device="cuda:0")
        from PIL import Image
        from stemseg_amd.inference.output_utils import MaskMaterializer
        keep2, masks = MaskMaterializer(-1).process_sequence((90, 120), idx, track, life, tuple(fg_p.shape[-2:]), 4.0, 10)
        assert keep == keep2 and len(keep) >= 2
        for t in range(12):
            png = np.asarray(Image.open(r"%s/out/results/seq0/%%05d.png" %% t))
            assert png.shape == (90, 120) and np.array_equal(png, masks[t].cpu().numpy())
        # registry / build_model level: the reference's own InferenceModel loop shape (inference_model.py:98-146)
        from stemseg.modeling.inference_model import preprocess_frames
        m = build_model(restore_pretrained_backbone_wts=False)
        m.load_state_dict(new)
        m = m.cuda()
        x, _ = preprocess_frames(frames[:8])
        feats = m.run_backbone(x)                                                 # {4, 8, 16, 32: [T, 256, h, w]}
        stacked = {s: f.permute(1, 0, 2, 3)[None].contiguous() for s, f in feats.items()}      # [1, C, T, h, w]
        out = m.embedding_head([stacked[s] for s in m.embedding_head_feature_map_scale]).squeeze(0)
        seed = m.seediness_head([stacked[s] for s in m.seediness_head_feature_map_scale]).squeeze(0)
        e = emb_a[0]
        assert float((out[:4] - e.embeddings).abs().max()) <= 1e-4 and float((seed - e.seediness).abs().max()) <= 1e-4
        assert float((out[4:6].exp() * 10 / e.bandwidths - 1).abs().max()) <= 1e-4
        print("OVERLAY-GPU-OK tracks kept", keep)
    """ % (textwrap.indent(IMPORT_LIST, "        ").strip(), tmp_path, tmp_path, tmp_path, tmp_path))
    assert "OVERLAY-GPU-OK" in out
