"""Sliding-window evaluators (host logic, runs on CPU and GPU alike) against G8 = the reference's
utils/sliding_evaluator.py / sliding_evaluator_plbl.py executed with the same stand-in network."""
import os

import numpy as np
import pytest
import torch

from mulactseg_amd import synth
from mulactseg_amd.utils.sliding_evaluator import SlidingEval, pad_margins, window_grid
from mulactseg_amd.utils.sliding_evaluator_plbl import SlidingEval as SlidingEvalFeat

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g8_sliding.npz")


@pytest.fixture(autouse=True)
def single_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("chunk", [1, 3])
def test_sliding_eval_matches_reference(tag, chunk):
    g = np.load(GOLD)
    seed, C, cls_n, H, W, crop = [int(v) for v in g[tag + '_cfg']]
    net = synth.tiny_window_net(seed, C, feat_dim=256)
    img = torch.from_numpy(np.random.RandomState(seed + 100).standard_normal((1, 3, H, W)).astype(np.float32))
    scores = SlidingEval(net, crop, 2 / 3, class_number=cls_n, windows_per_forward=chunk)(img)
    feats, scores2 = SlidingEvalFeat(net, crop, 2 / 3, class_number=cls_n, windows_per_forward=chunk)(img)
    assert tuple(scores.shape) == (cls_n, H, W) and tuple(feats.shape) == (256, H, W)
    assert torch.equal(scores, scores2)
    # same addends in the same order; the reference's float64 host array only widens the finished f32 sums
    tol = 1e-5 * float(np.abs(g[tag + '_scores']).max())
    assert float(np.abs(scores.numpy() - g[tag + '_scores']).max()) <= tol
    assert float(np.abs(feats.numpy()[::8, ::2, ::3] - g[tag + '_feats_sub']).max()) <= 1e-5 * float(np.abs(g[tag + '_feats_sub']).max())
    assert abs(float(feats.double().sum()) - float(g[tag + '_feats_sum'])) <= 1e-4 * max(1.0, abs(float(g[tag + '_feats_sum'])))
    assert np.array_equal(scores.numpy().argmax(0), g[tag + '_scores'].argmax(0))


def test_window_grid_cityscapes():
    """1024x2048, crop 800, stride 2/3 (trainer/eval_slide.py:62-66): 2 x 4 windows, last ones clamped to the border."""
    grid = window_grid(1024, 2048, (800, 800), 2 / 3)
    assert grid == [(0, 0), (0, 534), (0, 1068), (0, 1248), (224, 0), (224, 534), (224, 1068), (224, 1248)]
    assert pad_margins(1024, 2048, (800, 800)) == (0, 0, 0, 0)
    assert pad_margins(23, 61, (30, 30)) == (3, 4, 0, 0)


def test_small_image_is_padded_once():
    """Image smaller than the crop in both dims: one centred zero-padded window, cropped back (the reference's own
    branch for this case does not run -- 4-D indexing at sliding_evaluator.py:87 -- so this pins our reading of it)."""
    net = synth.tiny_window_net(5, 7, feat_dim=256)
    img = torch.from_numpy(np.random.RandomState(6).standard_normal((1, 3, 11, 14)).astype(np.float32))
    s = SlidingEval(net, 16, 2 / 3, class_number=7)(img)
    padded = torch.nn.functional.pad(img, (1, 1, 2, 3))
    ref = net(padded)[0][:, 2:13, 1:15]
    assert torch.equal(s, ref)


def test_slide_trainers_importable():
    import importlib
    for name in ("eval_slide", "active_slide", "eval_save_cosplbl_prop_includeonehot_slide"):
        m = importlib.import_module("mulactseg_amd.trainer." + name)
        assert hasattr(m, "ActiveTrainer")


def test_voc_variants_importable_and_parser_defaults():
    import importlib
    for name in ("base_voc", "active_voc", "eval_within_multihot_voc", "eval_save_cosplbl_prop_includeonehot_voc",
                 "eval_save_cosplbl_prop_includeonehot_voc_ms"):
        assert importlib.import_module("mulactseg_amd.trainer." + name)
    from mulactseg_amd.trainer import eval_within_multihot, eval_within_multihot_voc, eval_save_cosplbl_prop_includeonehot_voc as voc
    assert eval_within_multihot.ActiveTrainer.extra_channels == 1 and eval_within_multihot_voc.ActiveTrainer.extra_channels == 0
    assert voc.ActiveTrainer.extra_channels == 0 and voc.ActiveTrainer.include_onehot
    from mulactseg_amd.utils import common_voc
    a = common_voc.get_parser().parse_args([])
    assert a.num_classes == 21 and a.nseg == 32 and a.method == 'active_voc'


def test_voc_multiscale_ensemble_averages_scales_and_flips():
    """The ensemble of eval_save_cosplbl_prop_includeonehot_voc_ms on CPU tensors with the stand-in net: flipping the
    second half back and averaging equals the explicit computation."""
    import torch.nn.functional as F
    from mulactseg_amd.trainer import eval_save_cosplbl_prop_includeonehot_voc_ms as mod
    net = synth.tiny_window_net(3, 21, feat_dim=16)
    tr = object.__new__(mod.ActiveTrainer)
    tr.net, tr.device = net, torch.device('cpu')
    g = torch.Generator().manual_seed(0)
    base = torch.randn(3, 20, 28, generator=g)
    imgs = [F.interpolate(base[None], size=s, mode='bilinear', align_corners=False)[0] for s in ((10, 14), (20, 28), (30, 42))]
    lst = imgs + [im.flip(-1) for im in imgs]
    with torch.no_grad():
        feats, outs = tr.ensemble(lst, (20, 28))
        acc_f = acc_o = 0
        for k, im in enumerate(lst):
            f, o = net.feat_forward(im[None])
            if k >= 3:
                f, o = f.flip(-1), o.flip(-1)
            acc_f = acc_f + F.interpolate(f, size=(20, 28), mode='bilinear', align_corners=False)
            acc_o = acc_o + F.interpolate(o, size=(20, 28), mode='bilinear', align_corners=False)
    assert torch.allclose(outs, acc_o / 6, atol=1e-6) and torch.allclose(feats, F.normalize(acc_f / 6, dim=1), atol=1e-6)
    assert torch.allclose(feats.norm(dim=1), torch.ones(1, 20, 28), atol=1e-5)
