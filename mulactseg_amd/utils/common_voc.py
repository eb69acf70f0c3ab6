"""PASCAL VOC defaults of the argument parser -- reference ``utils/common_voc.py`` (a copy of ``utils/common.py`` whose
differences are defaults: :224-226 method / loader / active_method, :260-271 dataset names and lists, :312-328 21 classes,
batch 12, 30 k iterations, nseg 32)."""
from .common import AverageMeter, seed_everything  # noqa: F401
from . import common as _common


def get_parser():
    p = _common.get_parser()
    voc = dict(method='active_voc', loader='region_voc', active_method='random',
               src_dataset='voc', src_data_dir='./data/VOCdevkit',
               trg_dataset='voc', trg_data_dir='./data/VOCdevkit', trg_datalist='dataloader/init_data/voc/train_seed32.txt',
               region_dict='dataloader/init_data/voc/train_seed32.dict',
               val_dataset='voc', val_data_dir='./data/VOCdevkit', val_datalist='dataloader/init_data/voc/val.txt',
               num_classes=21, train_batch_size=12, val_batch_size=12, total_itrs=30000, nseg=32)
    known = {a.dest for a in p._actions}
    p.set_defaults(**{k: v for k, v in voc.items() if k in known})
    return p
