"""The selector plugins end to end on the GPU (real HipBackend through the C ABI): plugin surface of the
reference, expected values from the executed reference (g1, g2), bit-exact equality with the C oracle."""
import os
import pickle
import tempfile
import types

import numpy as np
import pytest
import torch

from helpers import FakePool, fake_trainer, selector_args
from test_oracle_golden import GOLDEN, g1_inputs, g2_inputs, tuples_to_arrays

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_pixbal_banignore_on_gpu_matches_reference_and_oracle():
    _need_gpu()
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd.active_selection.engine import HipBackend
    from mulactseg_amd.dataloader import RegionActiveDataset
    from helpers import OracleBackend
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    n_img, S = int(g['n_img']), int(g['S'])
    tmp = tempfile.mkdtemp()
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, model_save_dir=tmp, active_method='pixbal')
    sel = banignore.RegionSelector(args)
    tr = fake_trainer('cuda:0', tmp)
    scores, hist = sel.calculate_scores_tensor(tr, FakePool(z, spx, im_idx, suppix), want_hist=True)
    assert isinstance(sel.backend, HipBackend)                       # the native path ran, nothing else
    assert np.array_equal(hist.cpu().numpy(), g['region_ntop1'])
    assert np.allclose(scores.cpu().numpy(), g['scores_tensor'], rtol=RTOL, atol=1e-9)
    assert np.array_equal(scores.cpu().numpy() == 0, g['scores_tensor'] == 0)
    # bit-exact against the C oracle driven through the same host logic
    ref = banignore.RegionSelector(args)
    ref.backend = OracleBackend()
    rs, rh = ref.calculate_scores_tensor(fake_trainer(), FakePool(z, spx, im_idx, suppix), want_hist=True)
    assert np.array_equal(scores.cpu().numpy(), rs.numpy()) and np.array_equal(hist.cpu().numpy(), rh.numpy())
    assert np.array_equal(sel.cls_weight.cpu().numpy(), ref.cls_weight.numpy())
    # selection through the reference-shaped driver call
    pool = FakePool(z, spx, im_idx, suppix)
    pool.isselected = np.zeros((n_img, S), dtype=np.uint8)
    label = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=g['multi_hot'],
                                  id_to_index={"spx_%04d" % i: i for i in range(n_img)})
    active = RegionActiveDataset(args, pool, label)
    active.selection_iter = 1
    sel.select_next_batch(tr, active, int(g['budget']))
    active.wait_for_writes()                    # (the selection pickle is written by a background thread)
    with open(os.path.join(tmp, 'pixbal_selection_01.pkl'), 'rb') as f:
        consumed = pickle.load(f)
    cc, ci, cid = tuples_to_arrays(consumed, im_idx)
    assert np.array_equal(ci, g['consumed_img']) and np.array_equal(cid, g['consumed_id'])
    assert np.array_equal(pool.isselected, g['isselected'])


def test_voc_selectors_on_gpu_match_reference():
    _need_gpu()
    from mulactseg_amd.active_selection import my_bvsb, my_bvsb_predclsbal_pwr
    g = np.load(os.path.join(GOLDEN, "g2_voc.npz"))
    z, spx, im_idx, suppix = g2_inputs(g)
    S, C = int(g['S']), int(g['C'])
    tr = fake_trainer('cuda:0')
    for tag, method, ncls in (('plain', 'active_joint_multi_lossdecomp', C), ('strip', 'active_joint_multi_predignore_lossdecomp', C - 1)):
        args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, num_classes=ncls, method=method)
        sel = my_bvsb.RegionSelector(args)
        s = sel.calculate_scores_tensor(tr, FakePool(z, spx, im_idx, suppix)).cpu().numpy()
        assert np.allclose(s, g['bvsb_%s_scores_tensor' % tag], rtol=1e-4, atol=2e-6)
        tuples = sel.calculate_scores(tr, FakePool(z, spx, im_idx, suppix))
        _, si, sid = tuples_to_arrays(tuples, im_idx)
        assert np.array_equal(sid, g['bvsb_%s_list_id' % tag]) and np.array_equal(si, g['bvsb_%s_list_img' % tag])
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=S, num_classes=C, cls_weight_coeff=12.0,
                         method='active_joint_multi_lossdecomp')
    sel = my_bvsb_predclsbal_pwr.RegionSelector(args)
    scores, hist = sel.calculate_scores_tensor(tr, FakePool(z, spx, im_idx, suppix), want_hist=True)
    assert np.array_equal(hist.cpu().numpy(), g['pwr_region_ntop1'])
    assert np.allclose(scores.cpu().numpy(), g['pwr_scores_tensor'], rtol=RTOL, atol=1e-9)


def test_backend_refuses_cpu_device():
    _need_gpu()
    from mulactseg_amd import _lib
    from mulactseg_amd.active_selection.engine import HipBackend
    with pytest.raises(_lib.MulActSegHipError):
        HipBackend('cpu')


@pytest.mark.parametrize("tag,modname,method", [
    ('banignore', 'my_bvsb_banignore', 'active_joint_multi_predignore_lossdecomp'),
    ('clsbal_banignore', 'my_bvsb_clsbal_v2_banignore', 'active_joint_multi_predignore_lossdecomp'),
    ('clsbal', 'my_bvsb_clsbal_v2', 'active_joint_multi_lossdecomp')])
def test_remaining_selectors_on_gpu_match_reference(tag, modname, method):
    _need_gpu()
    import importlib
    from test_oracle_golden import g7_inputs
    g = np.load(os.path.join(GOLDEN, "g7_selectors.npz"))
    z, spx, im_idx, suppix = g7_inputs(g, tag)
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=int(g['S']), num_classes=int(g[tag + '_ncls']), method=method)
    sel = importlib.import_module("mulactseg_amd.active_selection." + modname).RegionSelector(args)
    s = sel.calculate_scores_tensor(fake_trainer('cuda:0'), FakePool(z, spx, im_idx, suppix)).cpu().numpy()
    ref = g[tag + '_scores_tensor']
    assert np.array_equal(s == 0, ref == 0)
    assert np.allclose(s, ref, rtol=1e-4, atol=2e-6)
    tuples = sel.calculate_scores(fake_trainer('cuda:0'), FakePool(z, spx, im_idx, suppix))
    _, si, sid = tuples_to_arrays(tuples, im_idx)
    assert np.array_equal(si, g[tag + '_list_img']) and np.array_equal(sid, g[tag + '_list_id'])


def test_pixbal_selector_under_an_initialised_rccl_group():
    """The sharded code path (ShardPlan, all-gather of class sums and scores over RCCL, 'nccl' backend) with a process
    group of one rank on the real device: same bits as the non-distributed run.  (N > 1 is covered under gloo in
    tests/test_distributed_cpu.py; this checks that the collectives accept the engine's device tensors.)"""
    _need_gpu()
    import torch.distributed as dist
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    tmp = tempfile.mkdtemp()
    args = selector_args(val_batch_size=int(g['batch_size']), nseg=int(g['S']), model_save_dir=tmp, active_method='pixbal')
    base = banignore.RegionSelector(args)
    s0, h0 = base.calculate_scores_tensor(fake_trainer('cuda:0', tmp), FakePool(z, spx, im_idx, suppix), want_hist=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sel = banignore.RegionSelector(args)
        s1, h1 = sel.calculate_scores_tensor(fake_trainer('cuda:0', tmp), FakePool(z, spx, im_idx, suppix), want_hist=True)
        assert torch.equal(s0, s1) and torch.equal(h0, h1)
        assert np.array_equal(base.cls_weight.cpu().numpy(), sel.cls_weight.cpu().numpy())
        for two_pass in (True,):
            args2 = selector_args(val_batch_size=int(g['batch_size']), nseg=int(g['S']), model_save_dir=tmp, active_method='pixbal',
                                  two_pass_scoring=two_pass)
            s2, _ = banignore.RegionSelector(args2).calculate_scores_tensor(fake_trainer('cuda:0', tmp), FakePool(z, spx, im_idx, suppix),
                                                                            want_hist=True)
            assert np.allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=1e-6, atol=1e-9)
    finally:
        dist.destroy_process_group()


def test_pool_batches_on_two_streams_give_the_same_bits():
    """Round 6: consecutive pool batches alternate between two HIP streams (RegionSelector._iterate, MAS_POOL_STREAMS): with a REAL
    model in the loop (seeded DeepLabv3+ on resident pictures, four batches, a short last one) scores, histograms and class weights
    equal the one-stream round bit for bit -- every accumulator of the scans is an integer sum or a per-picture row."""
    _need_gpu()
    from mulactseg_amd import synth
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    n_img, H, W, S, B = 7, 256, 512, 64, 2          # (16 x 32 planes at stride 16: every convolution on this package's kernels -- checked
    #                                                  below; narrower planes go to MIOpen, whose solver choice is not run-to-run stable)
    torch.manual_seed(3)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(5)
    pics = torch.randn((n_img, 3, H, W), generator=g, device=dev)
    maps = torch.from_numpy(np.stack([synth.superpixel_map(900 + i, H, W, S) for i in range(n_img)])).to(dev)

    class Pool(torch.utils.data.Dataset):
        device_resident = True
        im_idx = [["i/%03d.png" % i, "l/%03d.png" % i, "s/spx_%04d.pkl" % i] for i in range(n_img)]
        suppix = {k[2]: list(range(S)) for k in im_idx}

        def __len__(self):
            return n_img

        def __getitem__(self, i):
            return {'images': pics[i], 'spx': maps[i]}
    out = {}
    keep = os.environ.get("MAS_POOL_STREAMS")
    try:
        for n in ("1", "2", "3"):
            os.environ["MAS_POOL_STREAMS"] = n
            tmp = tempfile.mkdtemp()
            args = selector_args(val_batch_size=B, nseg=S, model_save_dir=tmp, active_method='pixbal')
            sel = banignore.RegionSelector(args)
            tr = types.SimpleNamespace(net=net, device=dev, model_save_dir=tmp, selection_iter=1)
            from mulactseg_amd.models import deeplab
            deeplab.path_report(reset=True)
            scores, hist = sel.calculate_scores_tensor(tr, Pool(), want_hist=True)
            torch.cuda.synchronize()
            paths = deeplab.path_report(reset=True)
            assert "miopen+bn" not in paths.get("conv_bn_act", {}), paths
            out[n] = (scores.cpu().numpy().copy(), hist.cpu().numpy().copy(), sel.cls_weight.cpu().numpy().copy())
    finally:
        if keep is None:
            os.environ.pop("MAS_POOL_STREAMS", None)
        else:
            os.environ["MAS_POOL_STREAMS"] = keep
    assert float(out["1"][0].max()) > 0 and out["1"][1].sum() == n_img * H * W
    for n in ("2", "3"):
        for a, b in zip(out["1"], out[n]):
            assert np.array_equal(a, b), n
