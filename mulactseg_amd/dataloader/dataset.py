"""Validation / evaluation sets -- the reference's ``dataloader/dataset.py:12-76`` (``CityscapesGTA5``) and ``:190-278`` (``VOC``):
full pictures through the deterministic resize, labels encoded to training ids.  Samples are made on the device
(``picture_store.PictureStore`` + ``device_transforms.DeviceResize``); see ``region_cityscapes.py``."""
import os

import numpy as np
import torch

from .constant import id_to_train_id, id_to_train_id_u8, train_id_to_color, voc_cmap
from .picture_store import PictureStore

_SPLITS = ('train', 'test', 'val', 'active-label', 'active-ulabel', 'custom-set', 'eval')


class _EvalSet(torch.utils.data.Dataset):
    device_resident = True

    def __init__(self, root, datalist, split='train', transform=None, return_spx=False, store=None):
        if split not in _SPLITS:
            raise ValueError("Invalid split %r: one of %s" % (split, ', '.join(_SPLITS)))
        if transform is None:
            raise NotImplementedError("a transform is required (dataloader.transform.get_val_transform)")
        self.root = os.path.expanduser(root)
        self.transform = transform
        self.split = split
        self.return_spx = return_spx
        self.store = store if store is not None else PictureStore()
        self.im_idx = self.read_list(datalist) if datalist is not None else []

    def sample_files(self, index):
        img, lbl, spx = self.im_idx[index]
        return [('rgb', img), ('map', lbl)] + ([('ids', spx)] if self.return_spx else [])

    def prefetch(self, indices):
        self.store.prefetch([f for i in indices for f in self.sample_files(i)])

    def encode_on_device(self, raw):
        return raw.long()

    def __getitem__(self, index):
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        maps = [self.store.labelmap(lbl_fname)] + ([self.store.idmap(spx_fname)] if self.return_spx else [])
        image, out = self.transform(self.store.picture(img_fname), maps)
        sample = {'images': image, 'labels': self.encode_on_device(out[0]), 'fnames': self.im_idx[index]}
        if self.return_spx:
            sample['spx'] = out[1]
        return sample

    def __len__(self):
        return len(self.im_idx)


class CityscapesGTA5(_EvalSet):
    """Datalist: three whitespace-separated paths per line, relative to the root (:36-42)."""
    _table = None

    def read_list(self, datalist):
        rows = np.atleast_2d(np.loadtxt(datalist, dtype='str'))
        return [[os.path.join(self.root, p) for p in row] for row in rows.tolist()]

    @classmethod
    def encode_target(cls, target):
        return id_to_train_id[np.array(target)]

    @classmethod
    def decode_target(cls, target):
        target[target == 255] = 19
        return train_id_to_color[target]

    def encode_on_device(self, raw):
        if self._table is None or self._table.device != raw.device:
            self._table = torch.from_numpy(id_to_train_id_u8).to(raw.device)
        return self._table[raw.long()].long()


class VOC(_EvalSet):
    """Datalist: bare picture names (:216-229); evaluation labels are ``VOC2012/SegmentationClass``."""
    cmap = voc_cmap()

    def __init__(self, root, datalist, split='train', transform=None, return_spx=False, dominant_labeling=False, store=None):
        self.dominant_labeling = dominant_labeling
        super().__init__(root, datalist, split, transform, return_spx, store=store)

    def read_list(self, datalist):
        from .region_voc import voc_paths
        names = np.atleast_1d(np.loadtxt(datalist, dtype='str')).tolist()
        dom = self.dominant_labeling and self.split not in ('test', 'val', 'eval')
        return [voc_paths(self.root, n, dom) for n in names]

    @classmethod
    def encode_target(cls, target):
        return np.array(target)

    @classmethod
    def decode_target(cls, target):
        target[target == 255] = 21
        return cls.cmap[target]
