"""SURVEY section 8(f)-3 end to end on the GPU: a tiny Cityscapes-shaped tree in the reference's on-disk formats (PNG pictures and
labels, pickled superpixel maps, region dictionary, datalist, multi-hot tensor) through ``get_active_dataset`` and the plugins --

* samples of the file-backed datasets against the Pillow restatement (oracle/augment.py) on the PIL-decoded files, same random draws;
* one acquisition round + two training steps through the file-backed path and through ``ResidentRegionDataset`` on the same tensors:
  identical selection, bit-identical losses;
* stage 2 (train_stage2_AL.py:21-51): ``eval_save_cosplbl_prop_includeonehot`` writes the pseudo-label PNGs from
  ``eval_region_cityscapes_all`` samples, ``region_cityscapes_plbl`` reads them back and ``active_predignore.ActiveTrainer.train`` takes
  a step whose loss is ``MyCrossEntropyLoss`` (utils/loss.py:10-21) of the same tensors."""
import logging
import os
import random
import types

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
H, W, NSEG, CROP = 128, 256, 64, 128        # (crops below 128 put 6 x 6 planes on MIOpen, ops.conv_train_plan: not run-to-run identical)
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _file_sets(tmp_path, extra=(), n=5):
    from mulactseg_amd import dataloader
    dataloader.register_dataset_factory(None)
    tree = helpers.write_cityscapes_tree(str(tmp_path / 'data'), n=n, H=H, W=W, nseg=NSEG)
    args = helpers.cityscapes_tree_args(tree, tmp_path / 'run', extra)
    os.makedirs(args.model_save_dir, exist_ok=True)
    aset = dataloader.get_active_dataset(args, train_transform=args.train_transform)
    aset.trg_pool_dataset.transform.target = (H, W)         # the pool at its native size (1024 x 2048 for real Cityscapes pictures)
    if hasattr(aset.trg_label_dataset.transform, 'scale_range') and not hasattr(aset.trg_label_dataset.transform, 'target'):
        aset.trg_label_dataset.transform.size = (CROP, CROP)
    return tree, args, aset


def test_samples_equal_the_pillow_pipeline_on_the_decoded_files(tmp_path):
    _gpu()
    from oracle import augment
    tree, args, aset = _file_sets(tmp_path)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    assert pool.device_resident and label.device_resident
    # pool sample: normalised picture, untouched id map, the picture's multi-hot table (region_cityscapes_or_tensor.py:47-52)
    item = pool[2]
    ref = tree['pictures'][2].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    ref = (ref - np.asarray(MEAN, np.float32)[:, None, None]) / np.asarray(STD, np.float32)[:, None, None]
    assert item['images'].is_cuda and np.array_equal(item['images'].cpu().numpy(), ref)
    assert item['spx'].dtype == torch.int64 and np.array_equal(item['spx'].cpu().numpy(), tree['spx'][2])
    assert np.array_equal(item['labels'].cpu().numpy(), tree['multi_hot'][2])
    # labelled sample after a selection: the augmentation of the reference on the decoded files, the mask = np.isin(spx, selected)
    aset.selection_iter = 1
    sel = [3, 7, 11, 40]
    order = [(1.0 - 0.01 * i, ','.join(pool.im_idx[1]), s) for i, s in enumerate(sel)]
    aset.expand_training_set(order, 10 ** 6, 'x')
    label.transform.rng = random.Random(11)
    twin = random.Random(11)
    for _ in range(3):                                                              # three draws: scales below and above 1
        s = label[0]
        p = augment.draw_params(twin, H, W, (CROP, CROP))
        img, (spx,) = augment.train_augment(tree['pictures'][1], [tree['spx'][1]], [NSEG], p, (CROP, CROP), MEAN, STD)
        assert np.array_equal(s['images'].cpu().numpy(), img) and np.array_equal(s['spx'].cpu().numpy(), spx)
        assert np.array_equal(s['spmask'].cpu().numpy(), np.isin(spx, sel)) and s['fnames'] == pool.im_idx[1]
        assert np.array_equal(s['labels'].cpu().numpy(), tree['multi_hot'][1])
    assert label.store.decodes == 4                                                 # every file was decoded once (2 pictures, 2 maps)
    # the validation set: raw label ids -> training ids on the device, pictures through the 1024 x 2048 resize
    from mulactseg_amd.dataloader import get_dataset
    val = get_dataset(args, name='cityscapes', data_root=args.val_data_dir, datalist=args.val_datalist, imageset='val')
    val.transform.target = (H, W)
    v = val[0]
    from PIL import Image
    raw = np.array(Image.open(val.im_idx[0][1]))
    assert np.array_equal(v['labels'].cpu().numpy(), val.encode_target(raw)) and v['labels'].dtype == torch.int64
    val.transform.target = (2 * H, 2 * W)
    v2 = val[0]
    up = augment.train_augment(np.array(Image.open(val.im_idx[0][0]).convert('RGB')), [raw], [255],
                               dict(scale=1, th=2 * H, tw=2 * W, gap_y=0, gap_x=0, i=0, j=0, flip=False), (2 * H, 2 * W), MEAN, STD)
    assert np.array_equal(v2['images'].cpu().numpy(), up[0]) and np.array_equal(v2['labels'].cpu().numpy(), val.encode_target(up[1][0]))


def _resident_twin(args, tree):
    from mulactseg_amd.dataloader import RegionActiveDataset
    from mulactseg_amd.dataloader.resident import ResidentRegionDataset
    pics = [torch.from_numpy(p).cuda() for p in tree['pictures']]
    spxs = [torch.from_numpy(s.astype(np.int16)).cuda() for s in tree['spx']]
    mh = torch.from_numpy(tree['multi_hot'])
    names = [tuple(os.path.join(tree['root'], p) for p in line.split('\t')) for line in tree['lines']]
    region = {n[2]: sorted(np.unique(s).tolist()) for n, s in zip(names, tree['spx'])}
    label = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-label')
    pool = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-ulabel', region_dict=region)
    label.transform.size = (CROP, CROP)
    return RegionActiveDataset(args, pool, label)


def _round(args, aset, tag):
    """One acquisition round (PixBal + ban-ignore on a seeded random-init model) and two training steps; returns the books."""
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as sel
    from mulactseg_amd.trainer import active_joint_multi_predignore_lossdecomp as T
    torch.manual_seed(0)
    random.seed(0)
    np.random.seed(0)
    tr = T.ActiveTrainer(args, logging.getLogger("test"), 1)
    aset.selection_iter = 1
    sel.RegionSelector(args).select_next_batch(tr, aset, args.active_selection_size)
    aset.wait_for_writes()
    seen = []
    add = tr._add_meters
    tr._add_meters = lambda keys, host: (seen.append(dict(zip(keys, host))), add(keys, host))[1]
    tr.train(aset)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    return {'label_im_idx': [list(k) for k in label.im_idx], 'label_suppix': {k: list(v) for k, v in label.suppix.items()},
            'pool_suppix': {k: list(v) for k, v in pool.suppix.items()}, 'losses': seen, 'trainer': tr}


def test_file_backed_round_equals_the_resident_round(tmp_path):
    """the same round through the files and through tensors that were resident from the start."""
    _gpu()
    tree, args, fset = _file_sets(tmp_path)
    a = _round(args, fset, 'file')
    b = _round(args, _resident_twin(args, tree), 'resident')
    assert a['label_im_idx'] == b['label_im_idx'] and a['label_suppix'] == b['label_suppix'] and a['pool_suppix'] == b['pool_suppix']
    assert sum(len(v) for v in a['label_suppix'].values()) > 5
    assert len(a['losses']) == 2 and a['losses'] == b['losses']                    # bit-identical: same kernels on the same bytes
    assert all(np.isfinite(v) for d in a['losses'] for v in d.values()) and any(d['train-loss'] > 0 for d in a['losses'])
    for f in ('datalist_01.pkl', 'checkpoint01.tar'):
        assert os.path.exists(os.path.join(args.model_save_dir, f)) or f == 'datalist_01.pkl'


def test_stage2_round_from_png_to_training_step(tmp_path):
    """train_AL.py round 1 -> eval_AL.py (pseudo-label PNGs) -> train_stage2_AL.py, all through files."""
    _gpu()
    from PIL import Image
    from mulactseg_amd import dataloader, ops
    from mulactseg_amd.trainer import active_predignore, eval_save_cosplbl_prop_includeonehot as G
    tree, args, fset = _file_sets(tmp_path, ['--active_selection_size', '160'])
    out = _round(args, fset, 'file')
    fset.dump_datalist()
    run = args.model_save_dir
    datalist, ckpt = os.path.join(run, 'datalist_01.pkl'), os.path.join(run, 'checkpoint01.tar')
    assert os.path.exists(datalist) and os.path.exists(ckpt)
    # -- eval_AL.py: --method eval_save_cosplbl_prop_includeonehot --loader eval_region_cityscapes_all --train_transform eval_spx
    a2 = helpers.cityscapes_tree_args(tree, run, ['--stage2', '--datalist_path', datalist, '--init_checkpoint', ckpt, '--resume_checkpoint', ckpt,
                                                  '--method', 'eval_save_cosplbl_prop_includeonehot', '--loader', 'eval_region_cityscapes_all',
                                                  '--train_transform', 'eval_spx', '--val_batch_size', '1'])
    a2.val_batch_size = 1
    set2 = dataloader.get_active_dataset(a2, train_transform=a2.train_transform)
    set2.trg_label_dataset.transform.target = (H, W)
    set2.selection_iter = 0
    set2.load_datalist(datalist)
    gen = G.ActiveTrainer(a2, logging.getLogger("test"), 0)
    gen.load_checkpoint(ckpt)
    table = gen.eval(set2, selection_iter=0)
    assert len(table.split(',')) == 1 + 20
    png_dir = os.path.join(run, 'plbl_gen', 'round_01')
    stems = sorted(k[0].split('/')[-1].split('_leftImg8bit')[0] for k in out['label_im_idx'])
    assert sorted(os.listdir(png_dir)) == [s + '.png' for s in stems]
    # a written PNG = the kernel output on the sample the file-backed evaluation set yields
    ds = set2.trg_label_dataset
    item = ds[0]
    assert set(item) == {'images', 'labels', 'target', 'spx', 'spmask', 'fnames'}
    k = tree['stems'].index(item['fnames'][0].split('/')[-1].split('_leftImg8bit')[0])
    want_lbl = np.where(tree['train_ids'][k] == 255, 19, tree['train_ids'][k])
    assert np.array_equal(item['labels'].cpu().numpy(), want_lbl)                   # ignore -> class 19 (eval_region_cityscapes_all.py:37-41)
    sel_ids = out['label_suppix'][item['fnames'][2]]
    assert np.array_equal(item['spmask'].cpu().numpy(), np.isin(tree['spx'][k], sel_ids))     # 'eval_save' in method: one-hot regions stay
    gen.net.eval()
    with torch.no_grad():
        feats, logits = gen.net.feat_forward_lowres(item['images'][None])
        want = ops.stage2_pseudo_labels(feats.contiguous(), logits.contiguous(), item['target'][None], item['spmask'][None], item['spx'][None], True)[0]
    got = np.array(Image.open(os.path.join(png_dir, tree['stems'][k] + '.png')))
    assert got.dtype == np.uint8 and got.shape == (H, W) and np.array_equal(got, want.cpu().numpy().astype(np.uint8))
    assert (got != 255).any()
    # -- train_stage2_AL.py: --method active_predignore --loader region_cityscapes_plbl --dominant_labeling --train_transform rescale_769_nospx
    a3 = helpers.cityscapes_tree_args(tree, run, ['--stage2', '--init_iteration', '1', '--datalist_path', datalist,
                                                  '--resume_checkpoint', os.path.join(run, 'checkpoint01.pkl'), '--init_checkpoint', ckpt,
                                                  '--method', 'active_predignore', '--loader', 'region_cityscapes_plbl', '--train_transform',
                                                  'rescale_769_nospx', '--loss_type', 'cross_entropy', '--finetune_itrs', '2', '--val_period', '2'])
    a3.or_labeling, a3.dominant_labeling, a3.fair_counting = False, True, False
    set3 = dataloader.get_active_dataset(a3, train_transform=a3.train_transform)
    set3.selection_iter = 1
    set3.load_datalist(datalist)
    train_set = set3.get_trainset()
    train_set.transform.size = (CROP, CROP)
    assert train_set.plbl_root == png_dir and len(train_set) == len(stems)
    # a sample = the Pillow pipeline on (decoded picture, decoded PNG)
    from oracle import augment
    train_set.transform.rng = random.Random(5)
    s = train_set[0]
    k = tree['stems'].index(s['fnames'][0].split('/')[-1].split('_leftImg8bit')[0])
    png = np.array(Image.open(os.path.join(png_dir, tree['stems'][k] + '.png')))
    p = augment.draw_params(random.Random(5), H, W, (CROP, CROP))
    img, (lab,) = augment.train_augment(tree['pictures'][k], [png], [255], p, (CROP, CROP), MEAN, STD)
    assert np.array_equal(s['images'].cpu().numpy(), img) and np.array_equal(s['labels'].cpu().numpy(), lab) and s['labels'].dtype == torch.int64
    tr = active_predignore.ActiveTrainer(a3, logging.getLogger("test"), 1)
    tr.load_checkpoint(ckpt)
    assert tr.target_dtype == torch.long and tr.net.classifier.proxy.shape[0] == 20
    seen = []
    loss_fun = tr.loss_fun

    def record(orig):
        def wrapped(*a, **kw):
            v = orig(*a, **kw)
            seen.append((a, v.detach().clone()))
            return v
        return wrapped
    loss_fun.forward = record(loss_fun.forward)
    if hasattr(loss_fun, 'forward_lowres'):
        loss_fun.forward_lowres = record(loss_fun.forward_lowres)
    before = [q.detach().clone() for q in tr.net.parameters()]
    fname = os.path.join(run, 'stage2_checkpoint01.tar')
    tr.train(set3, fname)
    assert len(seen) == 2 and os.path.exists(fname)
    with_labels = [sv for sv in seen if int((sv[0][-1] != 255).sum()) > 0]          # (pseudo labels are sparse: a crop may hold none)
    assert with_labels, "no crop of two steps met a pseudo label"
    (largs, value) = with_labels[0]
    if len(largs) == 3:                                                             # forward_lowres(quarter logits, size, labels)
        zq, size, labels = largs
        z = ops.upsample_bilinear(zq.detach(), (int(size[0]), int(size[1])))
    else:
        z, labels = largs
        z = z.detach()
    ref = torch.nn.functional.cross_entropy(z.double().cpu() / a3.ce_temp, labels.cpu(), ignore_index=255, reduction='mean')
    assert labels.dtype == torch.int64 and int((labels != 255).sum()) > 0
    assert abs(float(value) - float(ref)) <= 1e-4 * max(1.0, abs(float(ref)))      # MyCrossEntropyLoss (utils/loss.py:10-21), north-star tolerance
    moved = sum(float((q.detach() - b).abs().sum()) for q, b in zip(tr.net.parameters(), before))
    assert moved > 0 and np.isfinite(moved)


def test_voc_samples_equal_the_pillow_pipeline(tmp_path):
    """The VOC twins (region_voc_or_tensor / eval_region_voc_all / region_voc_plbl / dataset.VOC): JPEG pictures of different sizes,
    palette class PNGs, ``ExtResize(513) + ExtCenterCrop(513)`` for pool / evaluation samples, 513 x 513 random crops for training --
    every sample against the Pillow restatement on the PIL-decoded files."""
    _gpu()
    from PIL import Image
    from oracle import augment
    from mulactseg_amd import dataloader
    from mulactseg_amd.utils.common import get_parser
    dataloader.register_dataset_factory(None)
    tree = helpers.write_voc_tree(str(tmp_path / 'voc'), n=4)
    base = ['--src_dataset', 'voc', '--or_labeling', '--fair_counting', '--nseg', '150', '--num_classes', '21', '--trim_multihot_boundary',
            '--trim_kernel_size', '5', '--trg_data_dir', tree['root'], '--trg_datalist', tree['trg_datalist'], '--region_dict', tree['region_dict'],
            '--val_dataset', 'voc', '--val_data_dir', tree['root'], '--val_datalist', tree['val_datalist'], '-p', str(tmp_path / 'run')]
    a = get_parser().parse_args(base + ['--loader', 'region_voc_or_tensor', '--train_transform', 'rescale_513_multi_notrg'])
    os.makedirs(a.model_save_dir, exist_ok=True)
    aset = dataloader.get_active_dataset(a, train_transform=a.train_transform)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset

    def decoded(k):
        name = tree['names'][k]
        pic = np.array(Image.open(os.path.join(tree['root'], 'VOC2012/JPEGImages', name + '.jpg')).convert('RGB'))
        return pic, tree['spx'][k], tree['classes'][k]

    def centre(pic, maps, pads):
        H, W = pic.shape[:2]
        p, size = pool.transform.geometry(H, W)
        return augment.train_augment(pic, maps, pads, p, size, MEAN, STD)
    for k in (0, 1):                                    # a landscape and a portrait picture
        pic, spx, cls = decoded(k)
        s = pool[k]
        img, (m,) = centre(pic, [spx], [150])
        assert tuple(s['images'].shape) == (3, 513, 513) and np.array_equal(s['images'].cpu().numpy(), img)
        assert np.array_equal(s['spx'].cpu().numpy(), m) and tuple(s['labels'].shape) == (150, 21)
    # a labelled sample: 513 x 513 crop of a U(0.5, 2) rescale, pad id 150 never selected
    aset.selection_iter = 1
    ids = pool.suppix[pool.im_idx[2][2]][:7]
    aset.expand_training_set([(1.0 - 0.01 * i, ','.join(pool.im_idx[2]), s) for i, s in enumerate(ids)], 10 ** 6, 'x')
    label.transform.rng = random.Random(4)
    twin = random.Random(4)
    pic, spx, cls = decoded(2)
    for _ in range(2):
        s = label[0]
        p = augment.draw_params(twin, pic.shape[0], pic.shape[1], (513, 513))
        img, (m,) = augment.train_augment(pic, [spx], [150], p, (513, 513), MEAN, STD)
        assert np.array_equal(s['images'].cpu().numpy(), img) and np.array_equal(s['spx'].cpu().numpy(), m)
        assert np.array_equal(s['spmask'].cpu().numpy(), np.isin(m, ids)) and not bool(s['spmask'][s['spx'] == 150].any())
    # validation set (dataset.VOC): labels as stored, 255 = void
    val = dataloader.get_dataset(a, name='voc', data_root=a.val_data_dir, datalist=a.val_datalist, imageset='val')
    v = val[1]
    pic, spx, cls = decoded(1)
    img, (lab,) = centre(pic, [cls], [255])
    assert np.array_equal(v['images'].cpu().numpy(), img) and np.array_equal(v['labels'].cpu().numpy(), lab) and v['labels'].dtype == torch.int64
    # the stage-2 generator's view (eval_region_voc_all): void -> class 21, only selected regions that carry a class, the picture's (w, h)
    a2 = get_parser().parse_args(base + ['--loader', 'eval_region_voc_all', '--train_transform', 'eval_spx', '--method', 'eval_save_cosplbl_prop_includeonehot_voc'])
    set2 = dataloader.get_active_dataset(a2, train_transform=a2.train_transform)
    set2.trg_label_dataset.im_idx = [list(k) for k in label.im_idx]
    set2.trg_label_dataset.suppix = {k: list(v_) for k, v_ in label.suppix.items()}
    e = set2.trg_label_dataset[0]
    pic, spx, cls = decoded(2)
    img, (lab, m) = centre(pic, [cls, spx], [255, 150])
    assert np.array_equal(e['labels'].cpu().numpy(), np.where(lab == 255, 21, lab)) and np.array_equal(e['spx'].cpu().numpy(), m)
    has_cls = tree['multi_hot'][2, :, :21].sum(axis=1) != 0
    want = np.isin(m, [i for i in ids if has_cls[i]])
    assert np.array_equal(e['spmask'].cpu().numpy(), want) and e['imsizes'] == (pic.shape[1], pic.shape[0])
    assert tuple(e['target'].shape) == (150, 21)
