"""The production trainer plugin end to end on one GPU with a synthetic dataset factory: plugin surface of
the reference (ActiveTrainer(args, logger, selection_iter), .train(active_set), .eval(), checkpoints), the
first-step loss against the CPU restatement of the reference losses (oracle/port.py) on the same logits,
skip-on-zero-loss, and checkpoint round trip."""
import logging
import os
import tempfile
import types

import numpy as np
import pytest
import torch

from mulactseg_amd import synth

pytestmark = pytest.mark.gpu

N_CLS, S, H, W = 19, 64, 64, 96


class SynthTrain(torch.utils.data.Dataset):
    """Yields what region_cityscapes_or_tensor.py yields: images, multi-hot labels, spx (pad id S), spmask."""

    def __init__(self, n, empty_mask=False):
        self.n, self.empty = n, empty_mask
        self.selection_iter = 1

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rs = np.random.RandomState(100 + i)
        spx, msk = synth.train_crop(200 + i, H, W, S, frac_selected=0.3)
        if self.empty:
            msk[:] = False
        return {'images': torch.from_numpy(rs.standard_normal((3, H, W)).astype(np.float32)),
                'labels': torch.from_numpy(synth.multi_hot_targets(300 + i, S, N_CLS + 1)),
                'spx': torch.from_numpy(spx), 'spmask': torch.from_numpy(msk)}


class SynthVal(torch.utils.data.Dataset):
    def __len__(self):
        return 3

    def __getitem__(self, i):
        rs = np.random.RandomState(400 + i)
        lab = rs.randint(0, N_CLS, size=(H, W)).astype(np.int64)
        lab[rs.uniform(size=lab.shape) < 0.1] = 255
        return {'images': torch.from_numpy(rs.standard_normal((3, H, W)).astype(np.float32)), 'labels': torch.from_numpy(lab)}


def _args(tmp):
    from mulactseg_amd.utils.common import get_parser
    a = get_parser().parse_args([
        '-m', 'deeplabv3pluswn_resnet50deepstem', '--separable_conv', '--method', 'active_joint_multi_predignore_lossdecomp',
        '--ce_temp', '0.1', '--multi_ce_temp', '0.1', '--group_ce_temp', '0.1', '--coeff', '16.0', '--coeff_mc', '8.0',
        '--coeff_gm', '1.0', '--or_labeling', '--fair_counting', '--nseg', str(S), '--train_batch_size', '2',
        '--val_batch_size', '2', '--num_workers', '0', '--val_num_workers', '0', '--train_lr', '2e-5', '--finetune_itrs', '3',
        '--val_period', '2', '--log_period', '1', '-p', tmp])
    a.pretrained_backbone = False
    return a


def _trainer(tmp, empty=False):
    from mulactseg_amd import dataloader
    from mulactseg_amd.trainer import active_joint_multi_predignore_lossdecomp as T
    dataloader.register_dataset_factory(lambda args, name, data_root, datalist, imageset: SynthVal())
    torch.manual_seed(0)
    tr = T.ActiveTrainer(_args(tmp), logging.getLogger("test"), 1)
    active = types.SimpleNamespace(selection_iter=1, get_trainset=lambda: SynthTrain(4, empty))
    return tr, active


def test_production_trainer_step_matches_reference_loss():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import port
    tmp = tempfile.mkdtemp()
    tr, active = _trainer(tmp)
    assert tr.net.classifier.final.weight.shape[0] == N_CLS + 1            # predignore: one extra channel
    in_opt = {id(p) for g in tr.optimizer.param_groups for p in g['params']}
    assert in_opt == {id(p) for p in tr.net.parameters()}                  # every parameter is optimised, once
    assert tr.optimizer.param_groups[1]['lr'] == 10 * tr.optimizer.param_groups[0]['lr']      # head x10 (base.py:64-66)
    # reference value of the first step's loss: same batch, same dropout seed, losses restated on CPU
    tr.train_dataset_loader = tr.get_trainloader(active.get_trainset())
    torch.manual_seed(123)
    images, labels, spx, msk = tr._batch()
    tr.net.train()
    torch.manual_seed(7)
    with torch.no_grad():
        logits = tr.net(images).cpu()
    group = port.group_max_ce(logits, labels.cpu(), spx.cpu(), msk.cpu(), S, 0.1, 'onlymulti')
    ce, mc = port.merged_positive_ce(logits, labels.cpu(), spx.cpu(), msk.cpu(), 0.1, 'decomp')
    want = 16.0 * float(ce) + 8.0 * float(mc) + float(group)
    torch.manual_seed(7)
    g, c, m = tr.losses(tr.net(images), labels, spx, msk)
    got = 16.0 * float(c) + 8.0 * float(m) + float(g)
    assert abs(got - want) <= 1e-4 * max(1.0, abs(want))
    # full train() through the plugin surface: parameters move, checkpoint written on validation
    before = [p.detach().clone() for p in tr.net.parameters()]
    tr.train(active)
    moved = sum(float((p.detach() - q).abs().sum()) for p, q in zip(tr.net.parameters(), before))
    assert moved > 0 and np.isfinite(moved)
    ckpt = os.path.join(tmp, 'checkpoint01.tar')
    assert os.path.exists(ckpt)
    tr.load_checkpoint(ckpt)
    table = tr.eval(selection_iter=1)
    assert len(table.split(',')) == 1 + N_CLS + 1                           # mIoU, 19 classes, undefined-class IoU


def test_zero_loss_skips_the_optimizer_step():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    tmp = tempfile.mkdtemp()
    tr, active = _trainer(tmp, empty=True)
    tr.args.val_period = 1000
    before = [p.detach().clone() for p in tr.net.parameters()]
    tr.train(active)
    assert all(torch.equal(p.detach(), q) for p, q in zip(tr.net.parameters(), before))


def test_imagenet_checkpoint_loads_without_classifier():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    tmp = tempfile.mkdtemp()
    tr, _ = _trainer(tmp)
    from mulactseg_amd.models import get_model
    src = get_model('deeplabv3pluswn_resnet50deepstem', 21, 16, True, pretrained_backbone=False)   # other class count
    fname = os.path.join(tmp, 'resnet50_imagenet_pretrained.tar')
    torch.save({'model_state_dict': src.state_dict()}, fname)
    tr.load_checkpoint(fname)
    assert torch.equal(tr.net.backbone.conv1[0].weight.cpu(), src.backbone.conv1[0].weight)


class SynthLabelled(torch.utils.data.Dataset):
    """What the stage-2 loader yields per labelled image: image, GT labels, spx, spmask, multi-hot target, fnames."""

    def __init__(self, n):
        self.im_idx = [["img/%03d.png" % i, "gt/lbl_%03d.png" % i, "spx/%03d.pkl" % i] for i in reversed(range(n))]

    def __len__(self):
        return len(self.im_idx)

    def __getitem__(self, i):
        k = int(self.im_idx[i][0][4:7])
        rs = np.random.RandomState(500 + k)
        spx = synth.superpixel_map(600 + k, H, W, S)
        sel = rs.choice(S, size=8, replace=False)
        lab = rs.randint(0, N_CLS, size=(H, W)).astype(np.int64)
        return {'images': torch.from_numpy(rs.standard_normal((3, H, W)).astype(np.float32)), 'labels': torch.from_numpy(lab),
                'spx': torch.from_numpy(spx), 'spmask': torch.from_numpy(np.isin(spx, sel)),
                'target': torch.from_numpy(synth.multi_hot_targets(700 + k, S, N_CLS + 1)), 'fnames': self.im_idx[i]}


def test_stage2_generator_plugin_writes_pseudo_labels():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    from mulactseg_amd import dataloader, ops
    from mulactseg_amd.trainer import eval_save_cosplbl_prop_includeonehot as T
    tmp = tempfile.mkdtemp()
    a = _args(tmp)
    a.val_batch_size = 1
    a.init_checkpoint = os.path.join(tmp, 'checkpoint02.tar')
    a.plbl_type = None
    dataloader.register_dataset_factory(lambda *x, **k: SynthVal())
    torch.manual_seed(0)
    tr = T.ActiveTrainer(a, logging.getLogger("test"), 2)
    labelled = SynthLabelled(3)
    active = types.SimpleNamespace(trg_label_dataset=labelled, selection_iter=2)
    table = tr.eval(active, 2)
    assert len(table.split(',')) == 1 + N_CLS + 1
    out_dir = os.path.join(tmp, 'plbl_gen', 'round_02')
    files = sorted(os.listdir(out_dir))
    assert files == ['lbl_000.png', 'lbl_001.png', 'lbl_002.png']          # eval sorts im_idx (eval_within_multihot.py:34)
    # the PNG equals the kernel output on the same features
    item = labelled[[k[1] for k in labelled.im_idx].index('gt/lbl_001.png')]
    tr.net.eval()
    with torch.no_grad():
        feats, logits = tr.net.feat_forward_lowres(item['images'][None].cuda())
        want = ops.stage2_pseudo_labels(feats.contiguous(), logits.contiguous(), item['target'][None].cuda(),
                                        item['spmask'][None].cuda(), item['spx'][None].cuda(), True)[0].cpu().numpy()
    got = np.array(Image.open(os.path.join(out_dir, 'lbl_001.png')))
    assert got.dtype == np.uint8 and np.array_equal(got, want.astype(np.uint8))
    assert np.all(got[item['spmask'].numpy()] != 255)


def test_two_round_active_learning_loop_on_synthetic_data():
    """The reference's round structure (train_AL.py:37-85) end to end through the plugin names: random round, PixBal
    round (device scoring + selection), partial-label training, checkpoints, evaluation."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("al_demo", os.path.join(root, "examples", "train_AL_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tempfile.mkdtemp()
    hist = mod.main(["--rounds", "2", "--images", "5", "--iters", "4", "--budget", "30", "--out", out])
    assert [h[0] for h in hist] == [1, 2]
    assert hist[1][1] > hist[0][1] > 0                      # the labelled set grows every round
    # (the reference names every selection pickle after args.active_method, base.py:19,38)
    for f in ("checkpoint01.tar", "checkpoint02.tar", "datalist_01.pkl", "datalist_02.pkl",
              "my_bvsb_predclsbal_pwr_banignore_selection_01.pkl", "my_bvsb_predclsbal_pwr_banignore_selection_02.pkl"):
        assert os.path.exists(os.path.join(out, f)), f
    assert all(np.isfinite(h[2]) for h in hist)


def test_production_trainer_on_the_resident_data_path():
    """SURVEY 8f rank 4 end to end: pictures and superpixel maps resident in HBM, augmentation by mas_train_augment,
    batches from ResidentProvider (no DataLoader workers), the production trainer trains on them."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import random
    from mulactseg_amd.dataloader import ResidentProvider
    from mulactseg_amd.dataloader.resident import ResidentRegionDataset
    tmp = tempfile.mkdtemp()
    tr, _ = _trainer(tmp)
    rs = np.random.RandomState(0)
    n = 4
    pics = [torch.from_numpy(rs.randint(0, 256, size=(2 * H, 2 * W, 3)).astype(np.uint8)).cuda() for _ in range(n)]
    spxs = [torch.from_numpy(synth.superpixel_map(500 + i, 2 * H, 2 * W, S).astype(np.int16)).cuda() for i in range(n)]
    mh = torch.from_numpy(np.stack([synth.multi_hot_targets(600 + i, S, N_CLS + 1) for i in range(n)])).cuda()
    names = [("img%d" % i, "lbl%d" % i, "spx%d" % i) for i in range(n)]
    ds = ResidentRegionDataset(tr.args, pics, spxs, mh, names, split='active-label',
                               region_dict={"spx%d" % i: list(range(0, S, 3)) for i in range(n)}, rng=random.Random(5))
    ds.transform.size = (H, W)
    ds.selection_iter = 1
    active = types.SimpleNamespace(selection_iter=1, get_trainset=lambda: ds)
    before = [p.detach().clone() for p in tr.net.parameters()]
    tr.train(active)
    assert isinstance(tr.train_dataset_loader, ResidentProvider)
    moved = sum(float((p.detach() - q).abs().sum()) for p, q in zip(tr.net.parameters(), before))
    assert moved > 0 and np.isfinite(moved)


def test_five_round_loop_on_a_resident_pool_keeps_the_books(tmp_path):
    """BASELINE.json config 5 in miniature (examples/al_rounds_pool_scale.py, the script that produces
    profiles/r02/al_rounds_pool_scale.json at 2 975 x 2 048): five rounds on a resident pool; every round the labelled set grows by
    exactly the consumed prefix, the pool shrinks by the same regions, the budget walk stops where the reference's does and
    datalist_RR.pkl reloads -- the assertions live in the script and run at every size."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("al_pool", os.path.join(root, "examples", "al_rounds_pool_scale.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(["--rounds", "5", "--images", "24", "--height", "128", "--width", "256", "--nseg", "64", "--budget", "150", "--iters", "4",
                    "--crop", "96", "--val-images", "2", "--stage2-images", "2", "--out", str(tmp_path)])
    rs = rep["rounds"]
    assert [r["round"] for r in rs] == [1, 2, 3, 4, 5]
    tot = [r["labelled_regions_total"] for r in rs]
    assert all(b - a == r["regions_selected"] for a, b, r in zip([0] + tot, tot, rs))
    assert all(r["clicks"] > 150 for r in rs) and all(np.isfinite(r["val_miou_synthetic"]) for r in rs)
    assert rep["stage2_generation"]["pictures"] == 2
    for k in range(1, 6):
        assert os.path.exists(tmp_path / ("checkpoint%02d.tar" % k)) and os.path.exists(tmp_path / ("datalist_%02d.pkl" % k))


def test_five_round_loop_at_pool_scale(tmp_path):
    """BASELINE.json config 5 at its real size on one GPU (reference loop: train_AL.py:37-85): 2 975 pictures x 2 048 superpixels
    resident in HBM, 100 000 clicks per round, 5 rounds, 10 training iterations per round at the 768 crop.  Every round the
    script asserts its bookkeeping invariants (labelled set == previous + consumed prefix, no duplicates, pool shrinks by the same
    regions, budget walk stops at the reference's region, datalist_RR.pkl reloads) and -- for the four scored rounds -- that the
    prefix consumed by the device ordering + walk over ~6 M keys equals a numpy lexsort restatement of the reference's
    ``sorted(tuples, reverse=True)`` (active_selection/base.py:37) region for region."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.get_device_properties(0).total_memory < 80e9:
        pytest.skip("the resident pool needs ~35 GB of device memory")
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("al_pool_full", os.path.join(root, "examples", "al_rounds_pool_scale.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(["--rounds", "5", "--images", "2975", "--height", "1024", "--width", "2048", "--nseg", "2048", "--budget", "100000",
                    "--iters", "10", "--crop", "768", "--val-images", "4", "--stage2-images", "2", "--check-order", "--out", str(tmp_path)])
    rs = rep["rounds"]
    assert [r["round"] for r in rs] == [1, 2, 3, 4, 5]
    assert rep["pool"]["regions"] == 2975 * 2048
    tot = [r["labelled_regions_total"] for r in rs]
    assert all(b - a == r["regions_selected"] for a, b, r in zip([0] + tot, tot, rs))
    assert all(100000 < r["clicks"] <= 100004 for r in rs)
    assert all(r.get("order_checked_regions") == r["regions_selected"] for r in rs[1:])
    assert all(np.isfinite(r["val_miou_synthetic"]) for r in rs)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "al_rounds_pool_scale_test.json"), "w") as f:
        import json
        json.dump(rep, f, indent=1)
