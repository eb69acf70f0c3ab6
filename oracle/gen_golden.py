"""Generate ``tests/golden/*.npz`` by EXECUTING THE REFERENCE'S OWN PYTHON (build container only).

ORACLE / TEST INFRASTRUCTURE ONLY.  Run:  ``python oracle/gen_golden.py``  (needs ``/root/reference``).

The reference is imported from where it lies (``oracle/refshim/install.py`` supplies stand-ins for
its missing third-party imports); nothing of it is copied.  Inputs come from the seeded generators in
``mulactseg_amd/synth.py`` so the fixtures hold only seeds, tiny inputs' checksums and the
reference's OUTPUTS.  The fixtures are data: arrays in, arrays out.

Fixtures:
  g1_pixbal_city.npz   PixBal + ban-ignore scorer, Cityscapes-shaped (C=20), 3 images / batch 2,
                       selection walk with fair counting                         (rows a-1..a-6)
  g2_voc.npz           VOC-shaped plumbing: my_bvsb (with/without 'predignore') and
                       my_bvsb_predclsbal_pwr (C=21, S=150, odd image size)      (row a-5, config 1)
  g3_losses.npz        stage-1 losses fwd + dz for every in-scope loss class      (rows a-7, a-8, a-11)
  g5_miou.npz          MeanIoU / IoUIgnore counters incl. an unseen class         (row a-12)
  g10_train.npz        the network in TRAINING mode + two AdamW / PolyLR steps of the production objective
                       (batch statistics, running-stat update, shared proxy gradient, optimizer groups)   (rows a-9, a-10)
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
import install as refshim  # noqa: E402

from mulactseg_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


class CaptureLocals:
    """Grab the local variables of one reference function at its return (no source edits)."""

    def __init__(self, func_name, file_suffix):
        self.func_name, self.file_suffix, self.locals = func_name, file_suffix, None

    def __enter__(self):
        def prof(frame, event, arg):
            if event == 'return' and frame.f_code.co_name == self.func_name \
                    and frame.f_code.co_filename.endswith(self.file_suffix):
                self.locals = dict(frame.f_locals)
        sys.setprofile(prof)
        return self

    def __exit__(self, *a):
        sys.setprofile(None)


class FakePool(torch.utils.data.Dataset):
    """Pool dataset whose 'images' ARE the logits (the fake trainer's net is the identity)."""

    def __init__(self, logits, spx, im_idx, suppix):
        self.logits, self.spx, self.im_idx, self.suppix = logits, spx, im_idx, suppix

    def __len__(self):
        return len(self.im_idx)

    def __getitem__(self, i):
        return {'images': self.logits[i], 'spx': self.spx[i]}


def pool_inputs(seed, n_img, C, H, W, S, n_removed):
    z = synth.logits(seed, n_img, C, H, W)
    spx = np.stack([synth.superpixel_map(seed * 31 + i, H, W, S, n_missing=(1 if i == 1 else 0))
                    for i in range(n_img)])
    # datalist order is ascending in the path strings, like the reference's sorted datalist
    im_idx = [["leftImg8bit/train/c/img_%04d.png" % i, "gtFine/train/c/lbl_%04d.png" % i,
               "superpixel/train/c/spx_%04d.pkl" % i] for i in range(n_img)]
    rs = np.random.RandomState(seed + 5)
    suppix = {}
    for i in range(n_img):
        present = sorted(set(np.unique(spx[i]).tolist()))
        removed = set(rs.choice(present, size=n_removed, replace=False).tolist())
        suppix[im_idx[i][2]] = [s for s in present if s not in removed]
    return z, spx, im_idx, suppix


def tuples_to_arrays(tuples, im_idx):
    paths = {','.join(k): n for n, k in enumerate(im_idx)}
    return (np.array([t[0] for t in tuples], dtype=np.float64),
            np.array([paths[t[1]] for t in tuples], dtype=np.int64),
            np.array([t[2] for t in tuples], dtype=np.int64))


def load_region_active_dataset():
    path = os.path.join(refshim.REFERENCE_ROOT, "dataloader", "region_active_dataset.py")
    spec = importlib.util.spec_from_file_location("_ref_region_active_dataset", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.RegionActiveDataset


def gen_g1():
    from active_selection import my_bvsb_predclsbal_pwr_banignore as sel
    seed, n_img, C, H, W, S, bs = 11, 3, 20, 48, 64, 64, 2
    z, spx, im_idx, suppix = pool_inputs(seed, n_img, C, H, W, S, n_removed=5)
    # make "undefined" (last channel) dominant somewhere so that the ban fires
    z[0, C - 1, :16, :24] += 1.5
    args = types.SimpleNamespace(val_batch_size=bs, val_num_workers=0, nseg=S, active_method='x',
                                 num_classes=C - 1, ce_temp=0.1, cls_weight_coeff=6.0,
                                 method='active_joint_multi_predignore_lossdecomp', save_scores=False)
    pool = FakePool(torch.from_numpy(z), torch.from_numpy(spx), [list(k) for k in im_idx],
                    {k: list(v) for k, v in suppix.items()})
    trainer = types.SimpleNamespace(net=torch.nn.Identity(), device='cpu')
    selector = sel.RegionSelector(args)
    with CaptureLocals('calculate_scores', 'my_bvsb_predclsbal_pwr_banignore.py') as cap:
        scores = selector.calculate_scores(trainer, pool)
    loc = cap.locals
    sc, si, sid = tuples_to_arrays(scores, im_idx)

    # selection walk through the reference's own RegionActiveDataset.expand_training_set
    RegionActiveDataset = load_region_active_dataset()
    mh = np.stack([synth.multi_hot_targets(seed * 13 + i, S, C) for i in range(n_img)])
    label = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=mh,
                                  id_to_index={"spx_%04d" % i: i for i in range(n_img)})
    import tempfile
    tmp = tempfile.mkdtemp()
    aargs = types.SimpleNamespace(fair_counting=True, or_labeling=True, model_save_dir=tmp,
                                  finetune_itrs=1, wandb=types.SimpleNamespace(log=lambda *a, **k: None))
    pool.isselected = np.zeros((n_img, S), dtype=np.uint8)
    active = RegionActiveDataset(aargs, pool, label)
    active.selection_iter = 1
    budget = 40
    ordered = sorted(scores, reverse=True)            # active_selection/base.py:37
    active.expand_training_set(ordered, budget, 'pixbal')
    import pickle
    with open(os.path.join(tmp, 'pixbal_selection_01.pkl'), 'rb') as f:
        consumed = pickle.load(f)
    oc, oi, oid = tuples_to_arrays(ordered[:60], im_idx)
    cc, ci, cid = tuples_to_arrays(consumed, im_idx)
    np.savez_compressed(
        os.path.join(OUT, "g1_pixbal_city.npz"),
        seed=seed, n_img=n_img, C=C, H=H, W=W, S=S, batch_size=bs, ce_temp=0.1, coeff=6.0,
        n_removed=5, budget=budget, input_digest=digest(z, spx),
        cum=loc['cumulated_pred_prob'].numpy(), cls_weight=loc['cls_weight'].numpy(),
        region_ntop1=loc['top1nclasses'].view(n_img, S, C).numpy(),
        scores_tensor=loc['scores_tensor'].numpy(),
        list_score=sc, list_img=si, list_id=sid,
        sorted_score=oc, sorted_img=oi, sorted_id=oid,
        consumed_score=cc, consumed_img=ci, consumed_id=cid,
        isselected=pool.isselected, multi_hot=mh)
    print("g1: %d tuples, %d consumed, banned=%d" % (
        len(scores), len(consumed), int((loc['scores_tensor'] == 0).sum())))


def gen_g2():
    from active_selection import my_bvsb, my_bvsb_predclsbal_pwr
    seed, n_img, C, H, W, S, bs = 23, 3, 21, 33, 37, 150, 2
    z, spx, im_idx, suppix = pool_inputs(seed, n_img, C, H, W, S, n_removed=3)
    out = dict(seed=seed, n_img=n_img, C=C, H=H, W=W, S=S, batch_size=bs, ce_temp=0.1, coeff=12.0,
               n_removed=3, input_digest=digest(z, spx))
    trainer = types.SimpleNamespace(net=torch.nn.Identity(), device='cpu')

    def mk_pool():
        return FakePool(torch.from_numpy(z), torch.from_numpy(spx), [list(k) for k in im_idx],
                        {k: list(v) for k, v in suppix.items()})

    for tag, method in (('plain', 'active_joint_multi_lossdecomp'),
                        ('strip', 'active_joint_multi_predignore_lossdecomp')):
        args = types.SimpleNamespace(val_batch_size=bs, val_num_workers=0, nseg=S, active_method='x',
                                     num_classes=C, ce_temp=0.1, cls_weight_coeff=12.0,
                                     method=method, save_scores=False)
        with CaptureLocals('calculate_scores', 'my_bvsb.py') as cap:
            scores = my_bvsb.RegionSelector(args).calculate_scores(trainer, mk_pool())
        sc, si, sid = tuples_to_arrays(scores, im_idx)
        out['bvsb_%s_scores_tensor' % tag] = cap.locals['scores_tensor'].numpy()
        out['bvsb_%s_list_score' % tag] = sc
        out['bvsb_%s_list_img' % tag] = si
        out['bvsb_%s_list_id' % tag] = sid
    args = types.SimpleNamespace(val_batch_size=bs, val_num_workers=0, nseg=S, active_method='x',
                                 num_classes=C, ce_temp=0.1, cls_weight_coeff=12.0,
                                 method='active_joint_multi_lossdecomp', save_scores=False)
    with CaptureLocals('calculate_scores', 'my_bvsb_predclsbal_pwr.py') as cap:
        scores = my_bvsb_predclsbal_pwr.RegionSelector(args).calculate_scores(trainer, mk_pool())
    loc = cap.locals
    sc, si, sid = tuples_to_arrays(scores, im_idx)
    out.update(pwr_cum=loc['cumulated_pred_prob'].numpy(), pwr_cls_weight=loc['cls_weight'].numpy(),
               pwr_region_ntop1=loc['top1nclasses'].view(n_img, S, C).numpy(),
               pwr_scores_tensor=loc['scores_tensor'].numpy(),
               pwr_list_score=sc, pwr_list_img=si, pwr_list_id=sid)
    np.savez_compressed(os.path.join(OUT, "g2_voc.npz"), **out)
    print("g2: ok")


def gen_g7():
    """Remaining selectors: my_bvsb_banignore, my_bvsb_clsbal_v2, my_bvsb_clsbal_v2_banignore."""
    from active_selection import my_bvsb_banignore, my_bvsb_clsbal_v2, my_bvsb_clsbal_v2_banignore
    trainer = types.SimpleNamespace(net=torch.nn.Identity(), device='cpu')
    out = {}
    for tag, mod, C, ncls, method, seed in (
            ('banignore', my_bvsb_banignore, 20, 19, 'active_joint_multi_predignore_lossdecomp', 51),
            ('clsbal_banignore', my_bvsb_clsbal_v2_banignore, 20, 19, 'active_joint_multi_predignore_lossdecomp', 52),
            ('clsbal', my_bvsb_clsbal_v2, 21, 21, 'active_joint_multi_lossdecomp', 53)):
        n_img, H, W, S, bs = 3, 40, 56, 48, 2
        z, spx, im_idx, suppix = pool_inputs(seed, n_img, C, H, W, S, n_removed=4)
        if C == 20:
            z[0, C - 1, :14, :20] += 1.5
        args = types.SimpleNamespace(val_batch_size=bs, val_num_workers=0, nseg=S, active_method='x', num_classes=ncls,
                                     ce_temp=0.1, cls_weight_coeff=6.0, method=method, save_scores=False)
        pool = FakePool(torch.from_numpy(z), torch.from_numpy(spx), [list(k) for k in im_idx],
                        {k: list(v) for k, v in suppix.items()})
        with CaptureLocals('calculate_scores', mod.__name__.split('.')[-1] + '.py') as cap:
            scores = mod.RegionSelector(args).calculate_scores(trainer, pool)
        sc, si, sid = tuples_to_arrays(scores, im_idx)
        out.update({tag + '_seed': seed, tag + '_C': C, tag + '_ncls': ncls, tag + '_digest': digest(z, spx),
                    tag + '_scores_tensor': cap.locals['scores_tensor'].numpy(),
                    tag + '_list_score': sc, tag + '_list_img': si, tag + '_list_id': sid})
        if 'cls_weight' in cap.locals:
            out[tag + '_cls_weight'] = cap.locals['cls_weight'].numpy()
    np.savez_compressed(os.path.join(OUT, "g7_selectors.npz"), n_img=3, H=40, W=56, S=48, batch_size=2, **out)
    print("g7: ok")


def stage2_inputs(seed, N, C, Ch, H, W, S):
    """Stage-2 inputs: L2-normalised features [N,Ch,H,W] (as feat_forward yields them before upsampling), logits,
    multi-hot targets, superpixel map, selected mask (a few whole superpixels, one-hot and multi-hot)."""
    rs = np.random.RandomState(seed)
    cm = np.stack([synth.class_map(seed * 3 + i, H, W, C, blob=10) for i in range(N)])
    proto = rs.standard_normal((C, Ch)).astype(np.float32)
    feats = proto[cm].transpose(0, 3, 1, 2) + 0.6 * rs.standard_normal((N, Ch, H, W)).astype(np.float32)
    feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)
    z = synth.logits(seed + 1, N, C, H, W)
    spx = np.stack([synth.superpixel_map(seed * 5 + i, H, W, S) for i in range(N)])
    tgt = np.stack([synth.multi_hot_targets(seed * 7 + i, S, C, p_counts=(0.5, 0.3, 0.15, 0.05)) for i in range(N)])
    msk = np.zeros((N, H, W), dtype=bool)
    for i in range(N):
        chosen = rs.choice(S, size=max(3, S // 6), replace=False)
        msk[i] = np.isin(spx[i], chosen)
    msk[N - 1] = False                       # an image with nothing selected
    labels = rs.randint(0, C - 1, size=(N, H, W)).astype(np.int64)
    return feats, z, tgt, spx, msk, labels


def gen_g6():
    """Stage-2 cosine pseudo-label propagation (row a-13): trainer/eval_save_cosplbl_prop.py:121-314 and the
    production variant ..._includeonehot (valid = every selected pixel)."""
    import importlib
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda() at :266
    seed, N, C, Ch, H, W, S = 61, 3, 20, 16, 40, 56, 36
    feats, z, tgt, spx, msk, labels = stage2_inputs(seed, N, C, Ch, H, W, S)
    out = dict(seed=seed, N=N, C=C, Ch=Ch, H=H, W=W, S=S, input_digest=digest(feats, z, tgt, spx, msk))
    for tag, modname in (('multi', 'trainer.eval_save_cosplbl_prop'), ('all', 'trainer.eval_save_cosplbl_prop_includeonehot')):
        T = importlib.import_module(modname).ActiveTrainer
        tr = T.__new__(T)
        tr.args = types.SimpleNamespace(nseg=S, cosprop_threshold_method='median')
        tr.kernel = np.ones((3, 3), np.uint8)
        plbl = tr.pseudo_label_generation(torch.from_numpy(labels), torch.from_numpy(feats), torch.from_numpy(z),
                                          torch.from_numpy(tgt), torch.from_numpy(msk), torch.from_numpy(spx))
        out['plbl_' + tag] = plbl.numpy().astype(np.int16)
        print("g6 %s: labelled %d of %d pixels" % (tag, int((plbl != 255).sum()), plbl.numel()))
    np.savez_compressed(os.path.join(OUT, "g6_stage2.npz"), **out)


def loss_inputs(seed, N, C, H, W, S):
    z = synth.logits(seed, N, C, H, W)
    spx, msk = [], []
    for i in range(N):
        s, m = synth.train_crop(seed * 17 + i, H, W, S, frac_selected=0.25)
        spx.append(s)
        msk.append(m)
    spx, msk = np.stack(spx), np.stack(msk)
    tgt = np.stack([synth.multi_hot_targets(seed * 19 + i, S, C) for i in range(N)])
    # image 2: nothing selected (skip path); image 3: only one-hot regions selected
    msk[2] = False
    onehot = tgt[3].sum(axis=1) == 1
    msk[3] &= np.concatenate([onehot, [False]])[spx[3]]
    return z, tgt, spx, msk


def gen_g3():
    import utils.loss as L
    from trainer.active_joint_multi_predignore import MultiChoiceCE_, GroupMultiLabelCE_
    from trainer.active_joint_multi_predignore_lossdecomp import OnehotCEMultihotChoice
    from trainer.active_joint_multi_predignore_mclossablation2 import GroupMultiLabelCE_onlymulti
    seed, N, C, H, W, S, T = 37, 4, 20, 40, 44, 48, 0.1
    z, tgt, spx, msk = loss_inputs(seed, N, C, H, W, S)
    out = dict(seed=seed, N=N, C=C, H=H, W=W, S=S, temp=T, input_digest=digest(z, tgt, spx, msk))
    tt, ts, tm = torch.from_numpy(tgt), torch.from_numpy(spx), torch.from_numpy(msk)

    def run(tag, fn):
        zt = torch.from_numpy(z).clone().requires_grad_(True)
        res = fn(zt)
        res = res if isinstance(res, tuple) else (res,)
        for k, r in enumerate(res):
            out['%s_loss%d' % (tag, k)] = np.float32(float(r))
            if torch.is_tensor(r) and r.requires_grad:
                (g,) = torch.autograd.grad(r, zt, retain_graph=True)
                out['%s_grad%d' % (tag, k)] = g.numpy()
            else:
                out['%s_grad%d' % (tag, k)] = np.zeros_like(z)

    run('decomp', lambda zt: OnehotCEMultihotChoice(num_class=C - 1, temperature=T)(zt, tt, ts, tm))
    run('onlymulti', lambda zt: GroupMultiLabelCE_onlymulti(args=None, num_class=C - 1, num_superpixel=S,
                                                            temperature=T)(zt, tt, ts, tm))
    run('mc_predignore', lambda zt: MultiChoiceCE_(num_class=C - 1, temperature=T)(zt, tt, ts, tm))
    run('group_predignore', lambda zt: GroupMultiLabelCE_(args=None, num_class=C - 1, num_superpixel=S,
                                                          temperature=T)(zt, tt, ts, tm))
    # base classes (utils/loss.py) drop the last target column: C logits <-> C+1 target columns
    tgt_b = np.concatenate([tgt, np.zeros((N, S, 1), np.uint8)], axis=2)
    ttb = torch.from_numpy(tgt_b)
    run('mc_base', lambda zt: L.MultiChoiceCE(num_class=C, temperature=T)(zt, ttb, ts, tm))
    run('group_base', lambda zt: L.GroupMultiLabelCE(args=None, num_class=C, num_superpixel=S,
                                                     temperature=T)(zt, ttb, ts, tm))
    # production combination of train_impl: 16*ce + 8*mc + 1*group  (lossdecomp.py:102-104)
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    group = GroupMultiLabelCE_onlymulti(args=None, num_class=C - 1, num_superpixel=S, temperature=T)(zt, tt, ts, tm)
    ce, mc = OnehotCEMultihotChoice(num_class=C - 1, temperature=T)(zt, tt, ts, tm)
    total = 16.0 * ce + 8.0 * mc + 1.0 * group
    total.backward()
    out['total_loss'] = np.float32(float(total))
    out['total_grad'] = zt.grad.numpy()
    # stage-2 temperature CE (utils/loss.py:10-21)
    rs = np.random.RandomState(seed + 3)
    y = rs.randint(0, C, size=(N, H, W)).astype(np.int64)
    y[rs.uniform(size=y.shape) < 0.2] = 255
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    l2 = L.MyCrossEntropyLoss(ignore_index=255, temperature=T)(zt, torch.from_numpy(y))
    l2.backward()
    out['tce_loss'] = np.float32(float(l2))
    out['tce_grad'] = zt.grad.numpy()
    out['tce_digest'] = digest(y)
    np.savez_compressed(os.path.join(OUT, "g3_losses.npz"), **out)
    print("g3:", {k: float(v) for k, v in out.items() if k.endswith(('loss0', 'loss1', '_loss'))})


def gen_g4():
    """Model parity (row a-10): the reference's deeplabv3pluswn_resnet50deepstem + convert_to_separable_conv
    (models/__init__.py:46-49) with weights derived from the key names, eval mode."""
    from models.segmentation.modeling import deeplabv3pluswn_resnet50deepstem
    from models.segmentation import convert_to_separable_conv
    net = deeplabv3pluswn_resnet50deepstem(num_classes=20, output_stride=16, pretrained_backbone=False)
    convert_to_separable_conv(net.classifier)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synthetic_state_dict(shapes, seed=4)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    x = torch.from_numpy(np.random.RandomState(44).standard_normal(size=(1, 3, 129, 161)).astype(np.float32))
    with torch.no_grad():
        feats = net.backbone(x)
        quarter = net.classifier(feats)
        full = net(x)
        net.set_return_feat()                       # eval_within_multihot-style use (utils.py:13-15,28-34)
        feat_up, prob_up = net.feat_forward(x)
    keys = sorted(shapes)
    np.savez_compressed(os.path.join(OUT, "g4_model.npz"), seed=4, x_seed=44,
                        keys=np.array(keys), shapes=np.array([str(shapes[k]) for k in keys]),
                        low_level_mean=np.float64(feats['low_level'].double().mean()),
                        out_mean=np.float64(feats['out'].double().mean()),
                        quarter=quarter.numpy(), full_sub=full[:, :, ::3, ::3].numpy(),
                        feat_up_sub=feat_up[:, ::16, ::5, ::5].numpy(), prob_up_sub=prob_up[:, :, ::3, ::3].numpy(),
                        input_digest=digest(x.numpy()))
    print("g4: quarter", tuple(quarter.shape), "full", tuple(full.shape), float(quarter.abs().max()))


def gen_g5():
    from utils.miou import MeanIoU
    from utils.miou_evalignore import IoUIgnore
    seed, B, H, W, nc = 41, 2, 24, 32, 19
    rs = np.random.RandomState(seed)
    logits = rs.standard_normal(size=(2, B, nc + 1, H, W)).astype(np.float32)
    labels = rs.randint(0, nc, size=(2, B, H, W)).astype(np.int64)
    labels[labels == 7] = 3                     # class 7 never seen -> IoU counts as 100
    labels[rs.uniform(size=labels.shape) < 0.15] = 255
    m, g = MeanIoU(nc, 255), IoUIgnore(num_classes=nc, ignore_label=255)
    m._before_epoch()
    for step in range(2):
        p = torch.from_numpy(logits[step])
        t = torch.from_numpy(labels[step])
        m._after_step({'outputs': p[:, :-1].max(dim=1)[1], 'targets': t})   # predignore.py:200-203
        g._after_step({'outputs': p.max(dim=1)[1], 'targets': t})
    ious = m._after_epoch()
    np.savez_compressed(os.path.join(OUT, "g5_miou.npz"), seed=seed, B=B, H=H, W=W, nc=nc,
                        input_digest=digest(logits, labels),
                        seen=m.total_seen, correct=m.total_correct, positive=m.total_positive,
                        ious=np.array(ious, dtype=np.float64), miou=np.float64(np.mean(ious)),
                        ign=np.array([g.total_seen, g.total_correct, g.total_positive], dtype=np.float64),
                        ign_iou=np.float64(g._after_epoch()))
    print("g5: miou", np.mean(ious))


def gen_g8():
    """Sliding-window evaluators (BASELINE config 4, "sliding"): utils/sliding_evaluator.py and
    utils/sliding_evaluator_plbl.py on their working branch (image larger than the crop), with a two-convolution
    stand-in network.  Cases: both dims > crop with clamped last windows; one dim smaller than the crop (padding)."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    from utils.sliding_evaluator import SlidingEval as RefScores
    from utils.sliding_evaluator_plbl import SlidingEval as RefBoth
    out = {}
    for tag, seed, C, cls_n, H, W, crop in (('a', 81, 20, 19, 50, 77, 32), ('b', 82, 20, 20, 23, 61, 30), ('c', 83, 21, 21, 64, 64, 32)):
        net = synth.tiny_window_net(seed, C, feat_dim=256)
        img = torch.from_numpy(np.random.RandomState(seed + 100).standard_normal((1, 3, H, W)).astype(np.float32))
        with torch.no_grad():
            scores = RefScores(model=net, crop_size=crop, stride_rate=2 / 3, device='cpu', class_number=cls_n)(img)
            feats, scores2 = RefBoth(model=net, crop_size=crop, stride_rate=2 / 3, device='cpu', class_number=cls_n)(img)
        assert np.array_equal(scores, scores2)
        out[tag + '_cfg'] = np.array([seed, C, cls_n, H, W, crop])
        out[tag + '_scores'] = scores.astype(np.float32)
        out[tag + '_feats_sub'] = feats[::8, ::2, ::3].astype(np.float32)
        out[tag + '_feats_sum'] = np.float64(feats.sum())
        print("g8", tag, scores.shape, feats.shape, float(np.abs(scores).max()))
    np.savez_compressed(os.path.join(OUT, "g8_sliding.npz"), **out)


def _reference_ext_transforms():
    """The reference's own ``dataloader/ext_transforms.py``, imported from where it lies (by path: the package name
    ``dataloader`` is taken by the data-layer stand-in) with ``torchvision`` replaced by oracle/refshim/torchvision_shim."""
    import collections
    import collections.abc
    import importlib.util
    from oracle.refshim import torchvision_shim
    torchvision_shim.install_torchvision()
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable            # ext_transforms.py:568 (Python < 3.10 spelling)
    spec = importlib.util.spec_from_file_location("ref_ext_transforms", os.path.join(refshim.REFERENCE_ROOT, "dataloader", "ext_transforms.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_g9():
    """Training-time geometry produced by RUNNING the reference's transform classes -- ExtCompose([ExtRandomScale,
    ExtRandomCrop(pad_if_needed), ExtRandomHorizontalFlip, ExtToTensor, ExtNormalize]) as dataloader/transform.py:91-113 builds
    them -- on PIL images under a seeded ``random``: scale draw, pad, crop draws, flip draw, to-tensor and normalisation all come
    from /root/reference/dataloader/ext_transforms.py:172-192,443-520,323-341,384-437.  The draws the classes made are read back
    from the result (sizes) and cross-checked against oracle/augment.draw_params, the restatement the device data path uses."""
    import random
    from PIL import Image
    from oracle import augment
    et = _reference_ext_transforms()
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out = {}
    cases = []
    for k, (seed, H, W, crop, nseg) in enumerate(((1, 40, 64, (32, 32), 50), (2, 40, 64, (32, 32), 50), (3, 33, 47, (40, 40), 30),
                                                 (4, 40, 64, (32, 32), 50), (5, 24, 48, (24, 48), 20), (6, 61, 37, (32, 32), 64))):
        rs = np.random.RandomState(900 + seed)
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        lbl = rs.randint(0, 19, size=(H, W)).astype(np.uint8)
        spx = rs.randint(0, nseg, size=(H, W)).astype(np.int32)
        scale_range = (1.0, 1.0) if seed == 5 else (0.5, 2.0)
        tf = et.ExtCompose([et.ExtRandomScale(scale_range),
                            et.ExtRandomCrop(size=crop, pad_values=[255, nseg], padding=(124, 116, 104), pad_if_needed=True),
                            et.ExtRandomHorizontalFlip(),
                            et.ExtToTensor(dtype_list=['uint8', 'int']),
                            et.ExtNormalize(mean=mean, std=std)])
        random.seed(seed)
        t, (tl, ts) = tf(Image.fromarray(img), [Image.fromarray(lbl), Image.fromarray(spx).convert('I')])
        out['img_%d' % k] = t.numpy()
        out['lbl_%d' % k] = tl.numpy().astype(np.uint8)
        out['spx_%d' % k] = ts.numpy().astype(np.int64)
        # the same stream through the restated draw order: the parameters the device path will use for this seed
        p = augment.draw_params(random.Random(seed), H, W, crop, scale_range=scale_range)
        cases.append([seed, H, W, crop[0], crop[1], nseg, p['th'], p['tw'], p['gap_y'], p['gap_x'], p['i'], p['j'], int(p['flip'])])
        print("g9", cases[-1])
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "g9_augment.npz"), **out)


def sub256(a):
    """The fixture's deterministic cut of a tensor: every (numel // 256)-th element of the flattened array, at most 256."""
    a = np.ascontiguousarray(a).reshape(-1)
    return a[::max(1, a.size // 256)][:256].copy()


def train_inputs(seed, N, C, H, W, S):
    """Batch of the G10 training step: pictures, partial-label targets, superpixel maps, selection masks (all seeded)."""
    x = np.random.RandomState(seed).standard_normal(size=(N, 3, H, W)).astype(np.float32)
    spx, msk = zip(*[synth.train_crop(seed * 23 + i, H, W, S, frac_selected=0.3) for i in range(N)])
    tgt = np.stack([synth.multi_hot_targets(seed * 29 + i, S, C) for i in range(N)])
    return x, tgt, np.stack(spx), np.stack(msk)


def gen_g10():
    """TRAINING-mode pin of the network and of the optimizer (rows a-9, a-10): the reference's model builder
    (models/__init__.py:21-51: deeplabv3pluswn_resnet50deepstem + convert_to_separable_conv + set_bn_momentum(backbone, 0.1)) in
    .train(), Dropout probabilities set to 0 by attribute, two steps of the production objective
    (trainer/active_joint_multi_predignore_lossdecomp.py:83-116: net(images) upsampled, 16 ce + 8 mc + 1 group with the reference's
    loss classes) under the reference's own get_optim (trainer/base.py:64-69, AdamW with the classifier at cls_lr_scale x lr) and
    PolyLR (utils/scheduler.py:5-14).  Stored: logits and losses of both steps, a cut of every parameter gradient of step 1, every
    BatchNorm buffer after step 1 and after step 2, a cut of every parameter after step 2, the learning rates."""
    _gen_train("g10", 101, 4, 20, 129, 161, 48, full=True)


def gen_g11():
    """The WELL-CONDITIONED twin of G10: the same two steps on a [4,3,513,513] batch -- 33 x 33 maps at stride 16,
    so every BatchNorm of the deep layers averages over 4 x 1 089 = 4 356 samples per channel (G10: 4 x 99) and the f32 rounding of the
    convolutions is no longer amplified by noisy batch statistics.  The logit tensors are too large to store: kept are a strided cut
    (every third pixel of the quarter-resolution map, every twelfth of the full-resolution one) and the float64 norm of every
    (picture, class) plane of both, which pins what the cut does not sample."""
    _gen_train("g11", 211, 4, 20, 513, 513, 256, full=False)


def _gen_train(tag, seed, N, C, H, W, S, full):
    import models as ref_models
    from models.segmentation.modeling import deeplabv3pluswn_resnet50deepstem
    from models.segmentation import convert_to_separable_conv
    from trainer.base import BaseTrainer
    from utils.scheduler import PolyLR
    from trainer.active_joint_multi_predignore_lossdecomp import OnehotCEMultihotChoice
    from trainer.active_joint_multi_predignore_mclossablation2 import GroupMultiLabelCE_onlymulti
    T = 0.1
    net = deeplabv3pluswn_resnet50deepstem(num_classes=C, output_stride=16, pretrained_backbone=False)
    convert_to_separable_conv(net.classifier)
    ref_models.set_bn_momentum(net.backbone, momentum=0.1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synthetic_state_dict(shapes, seed=10)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    n_drop = 0
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
            n_drop += 1
    x, tgt, spx, msk = train_inputs(seed, N, C, H, W, S)
    xt, tt, ts, tm = torch.from_numpy(x), torch.from_numpy(tgt), torch.from_numpy(spx), torch.from_numpy(msk)
    tr = BaseTrainer.__new__(BaseTrainer)
    tr.args = types.SimpleNamespace(optimizer='adamw', cls_lr_scale=10.0, weight_decay=1e-5)
    tr.net = net
    tr.get_optim(my_lr=2e-5)
    sched = PolyLR(tr.optimizer, 10, power=0.9, min_lr=1e-6)
    group_fn = GroupMultiLabelCE_onlymulti(args=None, num_class=C - 1, num_superpixel=S, temperature=T)
    pos_fn = OnehotCEMultihotChoice(num_class=C - 1, temperature=T)
    quarter = {}
    hook = net.classifier.register_forward_hook(lambda m, i, o: quarter.__setitem__('q', o.detach().clone()))
    names = [n for n, _ in net.named_parameters()]
    out = dict(seed=seed, sd_seed=10, N=N, C=C, H=H, W=W, S=S, temp=T, lr=2e-5, cls_lr_scale=10.0, weight_decay=1e-5, max_iters=10,
               power=0.9, min_lr=1e-6, coeffs=np.array([16.0, 8.0, 1.0]), n_dropout=n_drop, input_digest=digest(x, tgt, spx, msk),
               param_names=np.array(names), buffer_names=np.array([n for n, _ in net.named_buffers()]),
               bn_momentum=np.array([m.momentum for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]))
    for step in (1, 2):
        tr.optimizer.zero_grad()
        preds = net(xt)
        group = group_fn(preds, tt, ts, tm)
        ce, mc = pos_fn(preds, tt, ts, tm)
        loss = 16.0 * ce + 8.0 * mc + 1.0 * group
        loss.backward()
        if full:
            out['quarter%d' % step] = quarter['q'].numpy()
            out['full_sub%d' % step] = preds.detach()[:, :, ::3, ::3].numpy()
        else:
            out['quarter_cut%d' % step] = quarter['q'][:, :, ::3, ::3].numpy().copy()
            out['quarter_norm%d' % step] = quarter['q'].double().flatten(2).norm(dim=2).numpy()
            out['full_cut%d' % step] = preds.detach()[:, :, ::12, ::12].numpy().copy()
            out['full_norm%d' % step] = preds.detach().double().flatten(2).norm(dim=2).numpy()
        out['losses%d' % step] = np.array([float(loss), float(ce), float(mc), float(group)], dtype=np.float32)
        if step == 1:
            for i, (n, p) in enumerate(net.named_parameters()):
                out['grad_%03d' % i] = sub256(p.grad.numpy())
                out['gnorm_%03d' % i] = np.float64(p.grad.double().norm())
        tr.optimizer.step()
        sched.step()
        out['lrs%d' % step] = np.array([g['lr'] for g in tr.optimizer.param_groups], dtype=np.float64)
        for i, (n, b) in enumerate(net.named_buffers()):
            out['buf%d_%03d' % (step, i)] = b.detach().numpy().copy()
    for i, (n, p) in enumerate(net.named_parameters()):
        out['param_%03d' % i] = sub256(p.detach().numpy())
    hook.remove()
    np.savez_compressed(os.path.join(OUT, "%s_train.npz" % tag), **out)
    print(tag + ": losses", out['losses1'], out['losses2'], "lrs", out['lrs1'], out['lrs2'], "params", len(names), "dropouts", n_drop)


if __name__ == "__main__":
    refshim.install()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)      # the goldens do not depend on it; keeps the run reproducible
    if len(sys.argv) > 1:         # python oracle/gen_golden.py g11 ... : only the named fixtures
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_g1()
    gen_g2()
    gen_g3()
    gen_g4()
    gen_g5()
    gen_g6()
    gen_g7()
    gen_g8()
    gen_g9()
    gen_g10()
    gen_g11()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
