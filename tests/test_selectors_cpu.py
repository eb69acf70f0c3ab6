"""Host logic of the selector plugins (plugin surface, sharding-free path, tuple lists, selection walk,
active-set bookkeeping) on CPU, with the oracle standing in for the GPU backend.  Expected values are
the golden vectors produced by executing the reference (tests/golden/g1, g2)."""
import os
import pickle
import tempfile
import types

import numpy as np
import pytest

from helpers import FakePool, OracleBackend, fake_trainer, selector_args
from test_oracle_golden import GOLDEN, g1_inputs, g2_inputs, tuples_to_arrays

RTOL = 1e-5      # floats: exact-arithmetic oracle vs the reference's ATen f32 (observed ~4e-7)


def _selector(modname, args):
    import importlib
    mod = importlib.import_module("mulactseg_amd.active_selection." + modname)
    sel = mod.RegionSelector(args)
    sel.backend = OracleBackend()
    return sel


def test_pixbal_banignore_plugin_matches_reference_g1():
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    n_img, S = int(g['n_img']), int(g['S'])
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=S)
    sel = _selector("my_bvsb_predclsbal_pwr_banignore", args)
    pool = FakePool(z, spx, im_idx, suppix)
    scores, hist = sel.calculate_scores_tensor(fake_trainer(), pool, want_hist=True)
    assert np.array_equal(hist.numpy(), g['region_ntop1'])                       # integer output: exact
    assert np.allclose(sel.cumulated_pred_prob, g['cum'], rtol=RTOL, atol=1e-8)
    assert np.allclose(sel.cls_weight.numpy(), g['cls_weight'], rtol=RTOL)
    ref = g['scores_tensor']
    assert np.array_equal(scores.numpy() == 0, ref == 0)                         # same banned / absent regions
    assert np.allclose(scores.numpy(), ref, rtol=RTOL, atol=1e-9)
    # drop-in tuple list API
    tuples = sel.calculate_scores(fake_trainer(), pool)
    sc, si, sid = tuples_to_arrays(tuples, im_idx)
    assert np.array_equal(si, g['list_img']) and np.array_equal(sid, g['list_id'])
    assert np.allclose(sc, g['list_score'], rtol=RTOL, atol=1e-9)
    # ordering identical to the reference's sorted(reverse=True)
    oc, oi, oid = tuples_to_arrays(sorted(tuples, reverse=True)[:60], im_idx)
    assert np.array_equal(oi, g['sorted_img']) and np.array_equal(oid, g['sorted_id'])


def _active_set(g, z, spx, im_idx, suppix, tmp):
    from mulactseg_amd.dataloader import RegionActiveDataset
    n_img, S = int(g['n_img']), int(g['S'])
    pool = FakePool(z, spx, im_idx, suppix)
    pool.isselected = np.zeros((n_img, S), dtype=np.uint8)
    label = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=g['multi_hot'],
                                  id_to_index={"spx_%04d" % i: i for i in range(n_img)})
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, model_save_dir=tmp)
    active = RegionActiveDataset(args, pool, label)
    active.selection_iter = 1
    return args, active


def test_select_next_batch_reproduces_reference_selection_g1():
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    tmp = tempfile.mkdtemp()
    args, active = _active_set(g, z, spx, im_idx, suppix, tmp)
    args.active_method = 'pixbal'
    sel = _selector("my_bvsb_predclsbal_pwr_banignore", args)
    n_pool_before = sum(len(v) for v in active.trg_pool_dataset.suppix.values())
    sel.select_next_batch(fake_trainer(save_dir=tmp), active, int(g['budget']))
    active.wait_for_writes()                    # (the selection pickle is written by a background thread)
    with open(os.path.join(tmp, 'pixbal_selection_01.pkl'), 'rb') as f:
        consumed = pickle.load(f)
    cc, ci, cid = tuples_to_arrays(consumed, im_idx)
    assert np.array_equal(ci, g['consumed_img']) and np.array_equal(cid, g['consumed_id'])      # bit-exact set+order
    assert np.allclose(cc, g['consumed_score'], rtol=RTOL)
    assert np.array_equal(active.trg_pool_dataset.isselected, g['isselected'])
    n_pool_after = sum(len(v) for v in active.trg_pool_dataset.suppix.values())
    assert n_pool_before - n_pool_after == len(consumed)
    lab = active.trg_label_dataset
    assert sum(len(v) for v in lab.suppix.values()) == len(consumed)
    # datalist round trip
    active.dump_datalist()
    pool_idx = list(active.trg_pool_dataset.im_idx)
    active.trg_pool_dataset.im_idx = []
    active.load_datalist()
    assert active.trg_pool_dataset.im_idx == pool_idx


def test_voc_selectors_match_reference_g2():
    g = np.load(os.path.join(GOLDEN, "g2_voc.npz"))
    z, spx, im_idx, suppix = g2_inputs(g)
    S, C = int(g['S']), int(g['C'])
    tr = fake_trainer()
    for tag, method, ncls, zz in (('plain', 'active_joint_multi_lossdecomp', C, z),
                                  ('strip', 'active_joint_multi_predignore_lossdecomp', C - 1, z)):
        args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, num_classes=ncls, method=method,
                             cls_weight_coeff=12.0)
        sel = _selector("my_bvsb", args)
        s = sel.calculate_scores_tensor(tr, FakePool(zz, spx, im_idx, suppix)).numpy()
        ref = g['bvsb_%s_scores_tensor' % tag]
        assert np.allclose(s, ref, rtol=1e-4, atol=2e-6)      # the (u-min)/max normalisation amplifies 1-ulp noise near 0
        tuples = sel.calculate_scores(tr, FakePool(zz, spx, im_idx, suppix))
        sc, si, sid = tuples_to_arrays(tuples, im_idx)
        assert np.array_equal(sid, g['bvsb_%s_list_id' % tag]) and np.array_equal(si, g['bvsb_%s_list_img' % tag])
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, num_classes=C, cls_weight_coeff=12.0,
                         method='active_joint_multi_lossdecomp')
    sel = _selector("my_bvsb_predclsbal_pwr", args)
    scores, hist = sel.calculate_scores_tensor(tr, FakePool(z, spx, im_idx, suppix), want_hist=True)
    assert np.array_equal(hist.numpy(), g['pwr_region_ntop1'])
    assert np.allclose(sel.cls_weight.numpy(), g['pwr_cls_weight'], rtol=RTOL)
    assert np.allclose(scores.numpy(), g['pwr_scores_tensor'], rtol=RTOL, atol=1e-9)


def test_random_and_dummy_selectors():
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    tmp = tempfile.mkdtemp()
    args, active = _active_set(g, z, spx, im_idx, suppix, tmp)
    args.active_method = 'my_random'
    from mulactseg_amd.active_selection import dummy, my_random
    import random
    random.seed(0)
    sel = my_random.RegionSelector(args)
    sel.select_next_batch(fake_trainer(save_dir=tmp), active, 25)
    with open(os.path.join(tmp, 'my_random_selection_01.pkl'), 'rb') as f:
        consumed = pickle.load(f)
    cost = sum(int(g['multi_hot'][int(p.split('spx_')[1][:4]), i].sum()) for _, p, i in consumed)
    assert cost > 25 and cost - int(g['multi_hot'][int(consumed[-1][1].split('spx_')[1][:4]), consumed[-1][2]].sum()) <= 25
    dummy.RegionSelector(args).select_next_batch(None, None, 0)


def test_class_weight_host_arithmetic_matches_c_oracle():
    from mulactseg_amd.active_selection.engine import class_weight_from_sums
    from oracle import exact
    rs = np.random.RandomState(0)
    n_img, C, hw = 7, 20, 1024 * 2048
    ps = (rs.uniform(0, 1, size=(n_img, C)) * hw * 2 ** 23 / C).astype(np.uint64)
    batch_of = (np.arange(n_img) // 4).astype(np.int32)
    cum_c, w_c = exact.class_weight(ps, hw, batch_of, 2, 6.0)
    cum_p, w_p = class_weight_from_sums(ps.view(np.int64), hw, batch_of, 2, 6.0)
    assert np.array_equal(cum_c, cum_p) and np.array_equal(w_c, w_p)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mulactseg_amd/ may import or load it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mulactseg_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ('from oracle', 'import oracle', 'libexact', 'oracle/port', 'oracle.port'):
                    assert needle not in text, (f, needle)


def test_reference_driver_imports_resolve_to_this_package():
    """train_AL.py:29-33 does importlib.import_module("active_selection." + name) / ("trainer." + name)."""
    import importlib
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import mulactseg_amd, importlib\n"
        "mulactseg_amd.install_aliases()\n"
        "m = importlib.import_module('active_selection.my_bvsb_predclsbal_pwr_banignore')\n"
        "assert m.RegionSelector.__module__.startswith('mulactseg_amd.'), m\n"
        "t = importlib.import_module('trainer.active_joint_multi_predignore_lossdecomp')\n"
        "assert hasattr(t, 'ActiveTrainer')\n"
        "from utils.loss import MultiChoiceCE, GroupMultiLabelCE, JointMultiLoss, MyCrossEntropyLoss\n"
        "from models import get_model\n"
        "from dataloader.utils import collate_fn, DataProvider\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stderr


def test_single_pass_and_two_pass_agree():
    """The single-pass scan (default) and the reference-structured two-pass scan give identical integers, identical
    class weights and scores within 2e-7 relative (only the per-pixel f32 rounding of bvsb*w differs)."""
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    S = int(g['S'])
    out = {}
    for two in (False, True):
        args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, two_pass_scoring=two)
        sel = _selector("my_bvsb_predclsbal_pwr_banignore", args)
        scores, hist = sel.calculate_scores_tensor(fake_trainer(), FakePool(z, spx, im_idx, suppix), want_hist=True)
        out[two] = (scores.numpy(), hist.numpy(), sel.cls_weight.numpy())
        assert sel._round.single_pass == (not two)
    assert np.array_equal(out[False][1], out[True][1])
    assert np.array_equal(out[False][2], out[True][2])
    a, b = out[False][0], out[True][0]
    assert np.array_equal(a == 0, b == 0)
    assert np.max(np.abs(a[b > 0] / b[b > 0] - 1)) < 2e-7
    assert np.array_equal(np.argsort(-a.ravel(), kind='stable'), np.argsort(-b.ravel(), kind='stable'))


@pytest.mark.parametrize("tag,modname,method", [
    ('banignore', 'my_bvsb_banignore', 'active_joint_multi_predignore_lossdecomp'),
    ('clsbal_banignore', 'my_bvsb_clsbal_v2_banignore', 'active_joint_multi_predignore_lossdecomp'),
    ('clsbal', 'my_bvsb_clsbal_v2', 'active_joint_multi_lossdecomp')])
def test_remaining_selectors_match_reference_g7(tag, modname, method):
    from test_oracle_golden import g7_inputs
    g = np.load(os.path.join(GOLDEN, "g7_selectors.npz"))
    z, spx, im_idx, suppix = g7_inputs(g, tag)
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=int(g['S']), num_classes=int(g[tag + '_ncls']), method=method)
    sel = _selector(modname, args)
    s = sel.calculate_scores_tensor(fake_trainer(), FakePool(z, spx, im_idx, suppix)).numpy()
    ref = g[tag + '_scores_tensor']
    assert np.array_equal(s == 0, ref == 0)
    assert np.allclose(s, ref, rtol=1e-4, atol=2e-6)
    if tag != 'banignore':
        assert np.allclose(sel.cls_weight.numpy(), g[tag + '_cls_weight'], rtol=2e-7)
    tuples = sel.calculate_scores(fake_trainer(), FakePool(z, spx, im_idx, suppix))
    sc, si, sid = tuples_to_arrays(tuples, im_idx)
    assert np.array_equal(si, g[tag + '_list_img']) and np.array_equal(sid, g[tag + '_list_id'])


def test_my_random_vectorised_selection_equals_the_tuple_sort(tmp_path):
    """my_random.select_next_batch (numpy lexsort over the same random.random() draws) consumes exactly the regions the
    reference path -- calculate_scores + sorted(reverse=True) + expand_training_set -- consumes, and leaves the global RNG
    in the same state."""
    import copy
    import pickle
    import random
    import types
    from mulactseg_amd.active_selection import base, my_random
    from mulactseg_amd.dataloader import RegionActiveDataset
    rs = np.random.RandomState(2)
    n, S, C = 7, 23, 20
    names = [["im/%02d.png" % i, "gt/%02d.png" % i, "sp/%02d.pkl" % i] for i in (3, 0, 6, 1, 5, 2, 4)]       # not in path order
    mh = (rs.rand(n, S, C) < 0.15).astype(np.uint8)
    mh[..., 1] = 1

    def sets(tag):
        d = tmp_path / tag
        d.mkdir()
        args = selector_args(nseg=S, model_save_dir=str(d), active_method='my_random')
        pool = types.SimpleNamespace(im_idx=copy.deepcopy(names), suppix={k[2]: [i for i in range(S) if (i + j) % 5] for j, k in enumerate(names)},
                                     isselected=np.zeros((n, S), np.uint8))
        label = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=mh, id_to_index={"%02d" % i: i for i in range(n)})
        a = RegionActiveDataset(args, pool, label)
        a.selection_iter = 1
        return args, a

    outs = []
    for tag, cls in (("vec", my_random.RegionSelector), ("ref", None)):
        args, aset = sets(tag)
        random.seed(11)
        sel = my_random.RegionSelector(args)
        if cls is None:
            base.RegionSelector.select_next_batch(sel, None, aset, 37)           # the generic tuple-sort path
        else:
            sel.select_next_batch(None, aset, 37)
        with open(tmp_path / tag / 'my_random_selection_01.pkl', 'rb') as f:
            outs.append((pickle.load(f), aset.trg_label_dataset.suppix, aset.trg_pool_dataset.suppix, random.random()))
    assert outs[0] == outs[1] and len(outs[0][0]) > 5


def test_pixbal_selector_scans_quarter_resolution_logits_when_the_model_offers_them():
    """A net with ``lowres_logits`` (models/deeplab.py) is asked for its quarter-resolution logits and the round goes through
    ``add_single_pass_lowres``; the scores equal those of the same selector on a net that upsamples first (oracle backend)."""
    import torch
    import torch.nn.functional as F
    from helpers import OracleBackend, fake_trainer
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd import synth
    from oracle import exact
    n_img, C, h, w, S = 5, 20, 6, 10, 24
    H, W = 4 * h, 4 * w
    zq = synth.logits(91, n_img, C, h, w)
    spx = np.stack([synth.superpixel_map(700 + i, H, W, S) for i in range(n_img)])
    im_idx = [["i/%03d.png" % i, "l/%03d.png" % i, "s/%03d.pkl" % i] for i in range(n_img)]
    suppix = {k[2]: list(range(S)) for k in im_idx}
    calls = []

    class LowNet(torch.nn.Module):
        lowres_logits = True

        def forward(self, x, lowres=False):
            calls.append(lowres)
            return x if lowres else torch.from_numpy(exact.upsample_bilinear(x.numpy(), H, W))

    class FullNet(torch.nn.Module):
        def forward(self, x):
            return torch.from_numpy(exact.upsample_bilinear(x.numpy(), H, W))

    out = []
    for net in (LowNet(), FullNet()):
        pool = FakePool(zq, spx, im_idx, suppix)
        sel = banignore.RegionSelector(selector_args(val_batch_size=2, nseg=S))
        sel.backend = OracleBackend()
        tr = fake_trainer()
        tr.net = net
        out.append(sel.calculate_scores_tensor(tr, pool).numpy())
    assert calls and all(calls)
    assert np.array_equal(out[0], out[1]) and float(out[0].max()) > 0
