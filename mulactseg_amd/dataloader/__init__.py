"""Data layer of the hot path: the tensor contract (shapes / dtypes), the active-set bookkeeping and the device data path.

``collate_fn`` / ``DataProvider`` (``utils.py``), ``RegionActiveDataset`` (``region_active_dataset.py``), readers of the
reference's on-disk formats (``formats.py``), the Pillow-exact device augmentation (``device_transforms.py``) and a region
dataset + batch provider for pictures resident in HBM (``resident.py``, ``utils.ResidentProvider``).  The reference's ~45
dataset variants and its image decoding are out of scope: plug a dataset in with ``register_dataset_factory``.
"""
from .region_active_dataset import RegionActiveDataset  # noqa: F401
from .utils import DataProvider, ResidentProvider, collate_fn  # noqa: F401


_DATASET_FACTORY = None


def register_dataset_factory(fn):
    """Plug a dataset constructor ``fn(args, name, data_root, datalist, imageset) -> Dataset`` in.

    The reference's dataset classes (``dataloader/region_*.py``, ``dataloader/dataset.py``) and their
    on-disk formats are outside the hot path; the trainers only need
    objects that yield ``{'images' f32[3,H,W], 'labels' ...}`` (+ ``'spx'``, ``'spmask'`` for stage 1)."""
    global _DATASET_FACTORY
    _DATASET_FACTORY = fn


def get_dataset(args, name, data_root, datalist, imageset):
    """Reference ``dataloader/__init__.py:get_dataset`` entry point, backed by the registered factory."""
    if _DATASET_FACTORY is None:
        raise NotImplementedError(
            "no dataset factory registered: call mulactseg_amd.dataloader.register_dataset_factory(fn); the "
            "reference's Cityscapes/VOC file readers are out of scope for the hot path")
    return _DATASET_FACTORY(args, name, data_root, datalist, imageset)


def get_slide_dataset(name, data_root, datalist, total_itrs=None, imageset='train'):
    """Reference ``dataloader/__init__.py:80-110``: the evaluation set at full resolution (Cityscapes resized to
    1024x2048, VOC 513 centre crop), no region annotations; the registered factory is asked for imageset
    ``'eval_slide'`` with ``args=None``."""
    assert imageset == "eval"
    assert name in ["cityscapes", "voc"]
    if _DATASET_FACTORY is None:
        raise NotImplementedError("no dataset factory registered: call mulactseg_amd.dataloader.register_dataset_factory(fn)")
    return _DATASET_FACTORY(None, name, data_root, datalist, 'eval_slide')
