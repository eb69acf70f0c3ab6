"""SURVEY section 8(f)-3 on the host: the file-backed datasets over the reference's on-disk formats (lists, dictionaries, label
tensors, decoders), ``get_active_dataset`` and the import surface the reference's drivers need after ``install_aliases()``.
Samples are made on the GPU (tests/test_filebacked_gpu.py)."""
import os
import pickle
import types

import numpy as np
import pytest

import helpers


def test_every_name_the_reference_drivers_import_resolves():
    """train_AL.py:14-16 / train_stage2_AL.py:12-14 / eval_AL.py: ``from dataloader import get_active_dataset``,
    ``from utils.common import initialization, get_parser, preprocess, arg_assert``, ``from utils.mylog import finalization,
    init_logging`` and the plugin modules the launch scripts name (script/open_source/*.sh)."""
    import importlib
    import mulactseg_amd
    mulactseg_amd.install_aliases()
    from dataloader import get_active_dataset                                       # noqa: F401
    from utils.common import arg_assert, get_parser, initialization, preprocess     # noqa: F401
    from utils.mylog import finalization, init_logging                              # noqa: F401
    for mod in ("dataloader.region_cityscapes_or_tensor", "dataloader.region_cityscapes_plbl", "dataloader.eval_region_cityscapes_all",
                "dataloader.region_voc_or_tensor", "dataloader.region_voc_plbl", "dataloader.eval_region_voc_all",
                "trainer.active_joint_multi_predignore_lossdecomp", "trainer.active_joint_multi_lossdecomp", "trainer.active_predignore",
                "trainer.active", "trainer.eval_save_cosplbl_prop_includeonehot", "trainer.eval_save_cosplbl_prop_includeonehot_voc_ms",
                "active_selection.my_bvsb_predclsbal_pwr_banignore", "active_selection.my_bvsb_predclsbal_pwr", "active_selection.my_random"):
        m = importlib.import_module(mod)
        assert m.__name__.startswith("mulactseg_amd."), mod
    for mod, cls in (("dataloader.region_cityscapes_or_tensor", "RegionCityscapesOr"), ("dataloader.region_cityscapes_plbl", "RegionCityscapes"),
                     ("dataloader.region_voc_or_tensor", "RegionVOCOr"), ("dataloader.region_voc_plbl", "RegionVOC")):
        assert hasattr(importlib.import_module(mod), cls)


def test_get_active_dataset_builds_the_two_splits_from_files(tmp_path):
    from mulactseg_amd.dataloader import RegionActiveDataset, get_active_dataset
    tree = helpers.write_cityscapes_tree(str(tmp_path / 'data'), n=4, H=32, W=48, nseg=16)
    args = helpers.cityscapes_tree_args(tree, tmp_path / 'run')
    aset = get_active_dataset(args, train_transform=args.train_transform)
    assert isinstance(aset, RegionActiveDataset)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    assert pool.split == 'active-ulabel' and label.split == 'active-label' and pool.store is label.store
    assert len(pool) == 4 and len(label) == 0 and label.suppix == {}
    # lists: absolute paths in file order, as region_cityscapes.py:70-76 builds them
    for k, line in enumerate(tree['lines']):
        assert pool.im_idx[k] == [os.path.join(tree['root'], p) for p in line.split('\t')]
        ids = pool.suppix[pool.im_idx[k][2]]
        assert ids == sorted(np.unique(tree['spx'][k]).tolist())                      # [nseg, [missing]] expanded
    assert np.array_equal(pool.multi_hot_cls.numpy(), tree['multi_hot']) and pool.isselected.shape == (4, 16)
    assert label.id_to_index == {s: k for k, s in enumerate(tree['stems'])}
    # the bookkeeping of a round on the file-backed lists, the datalist pickle and its reload
    aset.selection_iter = 1
    os.makedirs(args.model_save_dir, exist_ok=True)
    order = [(0.9, ','.join(pool.im_idx[1]), 3), (0.8, ','.join(pool.im_idx[2]), 5), (0.7, ','.join(pool.im_idx[1]), 4)]
    assert aset.expand_training_set(order, 10 ** 6, 'x') == 3
    assert label.im_idx == [pool.im_idx[1], pool.im_idx[2]] and label.suppix[pool.im_idx[1][2]] == [3, 4]
    assert 3 not in pool.suppix[pool.im_idx[1][2]] and pool.isselected[1, 3] == 1
    aset.dump_datalist()
    with open(os.path.join(args.model_save_dir, 'datalist_01.pkl'), 'rb') as f:
        data = pickle.load(f)
    assert data['trg_label_suppix'] == {pool.im_idx[1][2]: [3, 4], pool.im_idx[2][2]: [5]}
    again = get_active_dataset(args, train_transform=args.train_transform)
    again.load_datalist(os.path.join(args.model_save_dir, 'datalist_01.pkl'))
    assert again.trg_label_dataset.im_idx == label.im_idx and again.trg_pool_dataset.suppix == pool.suppix


def test_decoders_read_the_reference_formats(tmp_path):
    from PIL import Image
    from mulactseg_amd.dataloader import formats
    from mulactseg_amd.dataloader.picture_store import decode_map, decode_picture
    tree = helpers.write_cityscapes_tree(str(tmp_path), n=2, H=24, W=40, nseg=300, n_val=0)
    img, _, spx = [os.path.join(tree['root'], p) for p in tree['lines'][0].split('\t')]
    assert np.array_equal(decode_picture(img), tree['pictures'][0])
    ids = decode_map(spx, allow_u8=False)
    assert ids.dtype == np.int16 and np.array_equal(ids, tree['spx'][0]) and np.array_equal(formats.open_spx(spx), tree['spx'][0])
    lbl = os.path.join(tree['root'], 'gtFine/train/aachen/%s_gtFine_labelIds.png' % tree['stems'][0])
    assert decode_map(lbl).dtype == np.uint8 and np.array_equal(decode_map(lbl), tree['raw_labels'][0])
    big = np.arange(24 * 40, dtype=np.int32).reshape(24, 40) * 40                        # ids beyond int16
    with open(str(tmp_path / 'big.pkl'), 'wb') as f:
        pickle.dump({'labels': big}, f)
    assert decode_map(str(tmp_path / 'big.pkl'), allow_u8=False).dtype == np.int32
    # the palette picture of VOC class PNGs decodes to class indices, a grey JPEG picture to three equal channels
    pal = Image.fromarray((np.arange(24 * 40).reshape(24, 40) % 21).astype(np.uint8), mode='P')
    pal.putpalette([v % 256 for rgb in [(i, 2 * i, 3 * i) for i in range(256)] for v in rgb][:768])
    pal.save(str(tmp_path / 'cls.png'))
    assert np.array_equal(decode_map(str(tmp_path / 'cls.png')), np.arange(24 * 40).reshape(24, 40) % 21)
    Image.fromarray(tree['pictures'][0][..., 0]).save(str(tmp_path / 'grey.png'))
    g = decode_picture(str(tmp_path / 'grey.png'))
    assert g.shape == (24, 40, 3) and np.array_equal(g[..., 0], g[..., 2])


def test_label_tables_equal_the_cityscapes_definition():
    from mulactseg_amd.dataloader import constant
    from mulactseg_amd.dataloader.region_cityscapes import RegionCityscapes
    raw = np.arange(34, dtype=np.uint8).reshape(2, 17)
    enc = RegionCityscapes.encode_target(raw)
    want = np.full(34, 255)
    want[helpers.CITY_TRAIN_IDS] = np.arange(19)
    assert np.array_equal(enc.reshape(-1), want) and np.array_equal(constant.id_to_train_id_u8[:34], want)
    assert constant.id_to_train_id[-1] == 255 and (constant.id_to_train_id_u8[34:] == 255).all()
    assert constant.train_id_to_color.shape == (21, 3) and tuple(constant.train_id_to_color[13]) == (0, 0, 142)
    cm = constant.voc_cmap()
    assert tuple(cm[1]) == (128, 0, 0) and tuple(cm[6]) == (0, 128, 128) and tuple(cm[20]) == (0, 64, 128) and tuple(cm[255]) == (224, 224, 192)


def test_preprocess_and_arg_assert_follow_the_reference_rules(tmp_path):
    from mulactseg_amd.utils.common import arg_assert, get_parser, initialization, preprocess
    from mulactseg_amd.utils.mylog import finalization, init_logging, timediff
    import datetime
    a = get_parser().parse_args(['--nseg', '2048', '--or_labeling', '--method', 'active_joint_multi_predignore_lossdecomp',
                                 '--active_method', 'my_bvsb_predclsbal_pwr_banignore', '--coeff', '16.0', '--train_lr', '0.00002',
                                 '--finetune_itrs', '80000', '-p', str(tmp_path / 'city_mul_res50')])
    preprocess(a)
    assert a.trg_datalist == 'dataloader/init_data/cityscapes/train_seed2048_or.txt'
    assert a.model_save_dir == str(tmp_path / 'city_mul_res50') + ('_my_bvsb_predclsbal_pwr_banignore_sp2048_nlbl100.0k_iter80.0k_method-'
                                                                  'active_joint_multi_predignore_lossdecomp-_coeff16.0_ignFalse_lr2e-05_')
    assert a.session_name == 'active_joint_multi_predignore_lossdecomp_city_mul_res50'
    arg_assert(a)
    os.makedirs(a.model_save_dir)
    b = get_parser().parse_args(['--nseg_list', '128', '2048', '--dominant_labeling', '-p', str(tmp_path / 'city_mul_res50'), '--method', a.method,
                                 '--active_method', a.active_method, '--coeff', '16.0', '--train_lr', '0.00002', '--finetune_itrs', '80000',
                                 '--trg_datalist', 'lists/train_seed128.txt', '--region_dict', 'lists/train_seed128.dict'])
    preprocess(b)
    assert b.nseg == 2048                                                           # the largest of --nseg_list
    assert b.trg_datalist.endswith('train_seed2048_dominant.txt') and b.region_dict.endswith('train_seed2048.dict')
    assert b.model_save_dir == a.model_save_dir + '_1'                              # an existing run directory is never reused
    c = get_parser().parse_args(['--stage2', '-p', str(tmp_path / 'x'), '--datalist_path', '/r/datalist_01.pkl', '--resume_checkpoint', '/q/checkpoint01.pkl'])
    preprocess(c)
    assert c.model_save_dir == str(tmp_path / 'x')
    arg_assert(c)
    c.stage2 = False
    with pytest.raises(AssertionError):
        arg_assert(c)                                                                 # different run directories outside stage 2
    logger = initialization(types.SimpleNamespace(seed=3, model_save_dir=str(tmp_path / 'log')))
    assert os.path.isdir(tmp_path / 'log' / 'AL_record') and os.path.exists(tmp_path / 'log' / 'log_train.txt')
    args = types.SimpleNamespace(max_iterations=2, init_iteration=1, active_method='m', model_save_dir='d')
    init_logging(args)
    assert list(args.wandb_iou_table.columns) == ['round_v_miou', 'round-0', 'round-1', 'round-2']
    finalization(datetime.datetime.now(), {1: '1,2', 2: '3,4'}, logger, args)
    t0 = datetime.datetime(2024, 1, 1, 0, 0, 0)
    assert timediff(t0, t0 + datetime.timedelta(hours=3, minutes=4, seconds=5)) == '3h 4m 5s'


def test_transform_names_and_their_geometry():
    from mulactseg_amd.dataloader.transform import get_train_transform, get_train_transform_voc, get_val_transform
    a = types.SimpleNamespace(ignore_idx=255, nseg=2048, load_smaller_spx=False)
    t = get_train_transform(a, 'rescale_769_multi_notrg')
    assert t.size == (768, 768) and t.pad_values == [2048] and t.n_maps == 1 and tuple(t.fill) == (124, 116, 104)
    assert get_train_transform(a, 'rescale_769_nospx').pad_values == [255]
    assert get_train_transform(a, 'rescale_769').pad_values == [255, 2048]
    assert get_train_transform(a, None) is None
    with pytest.raises(NotImplementedError):
        get_train_transform(a, 'orig_notrg')
    v = get_train_transform_voc(types.SimpleNamespace(ignore_idx=255, nseg=150, load_smaller_spx=False), 'rescale_513_multi_notrg')
    assert v.size == (513, 513) and v.pad_values == [150]
    # ExtResize(513) + ExtCenterCrop(513) on a 375 x 500 (h x w) VOC picture: shorter side to 513, long side int(513 * 500 / 375) = 684
    p, size = get_val_transform('voc').geometry(375, 500)
    assert (p['th'], p['tw'], p['i'], p['j'], size) == (513, 684, 0, int(round((684 - 513) / 2.)), (513, 513))
    p, size = get_val_transform('voc').geometry(500, 333)
    assert (p['th'], p['tw'], p['i'], p['j']) == (int(513 * 500 / 333), 513, int(round((int(513 * 500 / 333) - 513) / 2.)), 0)
    p, size = get_val_transform('cityscapes').geometry(512, 1024)
    assert (p['th'], p['tw'], size) == (1024, 2048, (1024, 2048))


def test_voc_loaders_compose_the_paths_and_drop_the_undefined_column(tmp_path):
    """dataloader/region_voc.py:75-84 (three paths from a bare name), region_voc_or_tensor.py:30-62 (seeds_32 for --nseg 150, the last
    column of the label tensor dropped, rows by bare name)."""
    from mulactseg_amd.dataloader import get_active_dataset
    from mulactseg_amd.utils.common import get_parser
    tree = helpers.write_voc_tree(str(tmp_path / 'voc'), n=3)
    a = get_parser().parse_args(['--src_dataset', 'voc', '--loader', 'region_voc_or_tensor', '--train_transform', 'rescale_513_multi_notrg',
                                 '--or_labeling', '--fair_counting', '--nseg', '150', '--num_classes', '21', '--trim_multihot_boundary',
                                 '--trim_kernel_size', '5', '--trg_data_dir', tree['root'], '--trg_datalist', tree['trg_datalist'],
                                 '--region_dict', tree['region_dict'], '-p', str(tmp_path / 'run')])
    aset = get_active_dataset(a, train_transform=a.train_transform)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    assert type(pool).__name__ == 'RegionVOCOr' and len(pool) == 3 and len(label) == 0
    n0 = tree['names'][0]
    assert pool.im_idx[0] == [os.path.join(tree['root'], 'VOC2012/JPEGImages', n0 + '.jpg'),
                              os.path.join(tree['root'], 'VOC2012/SegmentationClass', n0 + '.png'),
                              os.path.join(tree['root'], 'superpixels/pascal_voc_seg/seeds_32/train/label', n0 + '.pkl')]
    assert pool.suppix[pool.im_idx[0][2]] == sorted(np.unique(tree['spx'][0]).tolist())
    assert tuple(pool.multi_hot_cls.shape) == (3, 150, 21) and np.array_equal(pool.multi_hot_cls.numpy(), tree['multi_hot'][:, :, :21])
    assert pool.id_to_index == {n: k for k, n in enumerate(tree['names'])}
    assert label.transform.size == (513, 513) and label.transform.pad_values == [150]
    assert pool.transform.geometry(120, 160)[1] == (513, 513)
    assert pool.encode_target(np.array([[3, 255]], dtype=np.uint8)).tolist() == [[3, 255]]
