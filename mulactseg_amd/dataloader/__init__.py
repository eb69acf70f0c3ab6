"""Data layer of the hot path: the tensor contract (shapes / dtypes), the active-set bookkeeping, the device data path and the
file-backed datasets over the reference's on-disk formats.

``collate_fn`` / ``DataProvider`` (``utils.py``), ``RegionActiveDataset`` (``region_active_dataset.py``), readers of the reference's
on-disk formats (``formats.py``), the Pillow-exact device augmentation (``device_transforms.py``, ``transform.py``), datasets whose
pictures are resident in HBM from the start (``resident.py``) or decoded from files on first use (``picture_store.py``;
``region_cityscapes*.py``, ``region_voc*.py``, ``eval_region_*_all.py``, ``dataset.py`` -- the loaders the reference's launch scripts
name), and ``get_active_dataset`` / ``get_dataset`` / ``get_slide_dataset`` with the reference's signatures
(``dataloader/__init__.py:9-186``).  The reference's ~40 ablation dataset variants are out of scope; any other dataset plugs in
with ``register_dataset_factory``.
"""
import importlib

from .region_active_dataset import RegionActiveDataset  # noqa: F401
from .transform import get_train_transform, get_train_transform_voc, get_val_transform  # noqa: F401
from .utils import DataProvider, ResidentProvider, collate_fn  # noqa: F401


_DATASET_FACTORY = None


def register_dataset_factory(fn):
    """Plug a dataset constructor ``fn(args, name, data_root, datalist, imageset) -> Dataset`` in: it then answers ``get_dataset`` /
    ``get_slide_dataset`` instead of the file-backed classes (synthetic data, other datasets).  ``None`` removes it.  The trainers
    only need objects that yield ``{'images' f32[3,H,W], 'labels' ...}`` (+ ``'spx'``, ``'spmask'`` for stage 1)."""
    global _DATASET_FACTORY
    _DATASET_FACTORY = fn


def get_dataset(args, name, data_root, datalist, total_itrs=None, imageset='train'):
    """Reference ``dataloader/__init__.py:9-78``: the validation / evaluation set (Cityscapes resized to 1024x2048, VOC resized to
    513 and centre-cropped), labels as training ids."""
    if _DATASET_FACTORY is not None:
        return _DATASET_FACTORY(args, name, data_root, datalist, imageset)
    assert imageset in ["val", "eval"]
    assert name in ["cityscapes", "voc"]
    from . import dataset as _ds
    transform = get_val_transform(name, n_maps=1, ignore_idx=getattr(args, 'ignore_idx', 255), nseg=getattr(args, 'nseg', 2048))
    if name == "cityscapes":
        return _ds.CityscapesGTA5(data_root, datalist, imageset, transform=transform)
    return _ds.VOC(data_root, datalist, imageset, transform=transform, dominant_labeling=getattr(args, 'dominant_labeling', False))


def get_slide_dataset(name, data_root, datalist, total_itrs=None, imageset='train'):
    """Reference ``dataloader/__init__.py:80-110``: the evaluation set at full resolution for the sliding-window evaluator."""
    assert imageset == "eval"
    assert name in ["cityscapes", "voc"]
    if _DATASET_FACTORY is not None:
        return _DATASET_FACTORY(None, name, data_root, datalist, 'eval_slide')
    from . import dataset as _ds
    cls = _ds.CityscapesGTA5 if name == "cityscapes" else _ds.VOC
    return cls(data_root, datalist, imageset, transform=get_val_transform(name))


def get_active_dataset(args, train_transform=None):
    """Reference ``dataloader/__init__.py:112-186``: the labelled set (training transform, starts empty) and the pool (the resize
    transform, every superpixel of ``args.region_dict``) of ``args.loader``, wrapped in ``RegionActiveDataset``.  The two datasets
    share one ``PictureStore``: a picture decoded for the acquisition pass is not decoded again for training."""
    voc = args.src_dataset == 'voc'
    if not voc and args.src_dataset != 'cityscapes':
        raise NotImplementedError("src_dataset %r (the reference's GTA5 / SYNTHIA choices have no region loader either)" % args.src_dataset)
    if getattr(args, 'active_mode', 'region') != 'region':
        raise NotImplementedError("active_mode %r" % args.active_mode)
    lbl_transform = (get_train_transform_voc if voc else get_train_transform)(args, train_transform)
    pool_transform = get_val_transform('voc' if voc else 'cityscapes', ignore_idx=args.ignore_idx, nseg=args.nseg)
    loader = importlib.import_module("%s.%s" % (__name__, args.loader.lower()))
    from .picture_store import PictureStore
    store = PictureStore()
    dominant = bool(getattr(args, 'dominant_labeling', False))
    if getattr(args, 'or_labeling', False):
        cls = loader.RegionVOCOr if voc else loader.RegionCityscapesOr
        label = cls(args, args.trg_data_dir, None, split='active-label', transform=lbl_transform, region_dict=args.region_dict,
                    dominant_labeling=dominant, loading=getattr(args, 'loading', 'binary'),
                    load_smaller_spx=getattr(args, 'load_smaller_spx', False), store=store)
        pool = cls(args, args.trg_data_dir, args.trg_datalist, region_dict=args.region_dict, split='active-ulabel', transform=pool_transform,
                   return_spx=True, store=store)
    else:
        cls = loader.RegionVOC if voc else loader.RegionCityscapes
        label = cls(args, args.trg_data_dir, None, split='active-label', transform=lbl_transform, region_dict=args.region_dict,
                    dominant_labeling=dominant, store=store)
        pool = cls(args, args.trg_data_dir, args.trg_datalist, region_dict=args.region_dict, split='active-ulabel', transform=pool_transform,
                   return_spx=True, dominant_labeling=dominant, store=store)
    if 'mseg' in args.loader.lower():
        raise NotImplementedError("the multi-segmentation loaders are outside the hot path")
    return RegionActiveDataset(args, pool, label)
