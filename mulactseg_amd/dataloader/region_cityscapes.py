"""File-backed Cityscapes region dataset -- the reference's ``dataloader/region_cityscapes.py:15-153`` (``RegionCityscapes``): same
constructor, ``im_idx`` / ``suppix`` bookkeeping, sample dictionary and class methods, so ``get_active_dataset`` /
``RegionActiveDataset`` / the trainers use it as they use the reference's.

What differs is WHERE a sample is made.  The reference decodes and augments PIL images in DataLoader workers and ships float crops to
the GPU; here the decoded picture and maps live in HBM (``picture_store.PictureStore``: decoded once by host threads) and one kernel
per sample (``device_transforms.DeviceTrainAugment`` / ``DeviceResize``, ``csrc/augment.hip`` -- bit-identical to the Pillow chain,
tests/golden G9) writes the normalised crop and the cropped maps.  Label encoding (``id_to_train_id``) and the "keep only the selected
superpixels" mask (``np.isin`` on the host in the reference, :116-127) are table lookups on the device.  ``device_resident = True``
tells the trainers / selectors to batch it without worker processes (``dataloader.utils.ResidentProvider``).
"""
import os

import numpy as np
import torch

from . import formats
from .constant import id_to_train_id, id_to_train_id_u8, train_id_to_color
from .picture_store import PictureStore

_SPLITS = ('train', 'test', 'val', 'active-label', 'active-ulabel', 'custom-set')


class RegionCityscapes(torch.utils.data.Dataset):
    device_resident = True
    default_region_dict = "dataloader/init_data/cityscapes/train.dict"

    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, store=None):
        if not hasattr(args, "prob_dominant"):
            args.prob_dominant = False
        if split not in _SPLITS:
            raise ValueError("Invalid split %r: one of %s" % (split, ', '.join(_SPLITS)))
        if transform is None:
            raise NotImplementedError("a transform is required (dataloader.transform.get_train_transform / get_val_transform)")
        self.args = args
        self.root = os.path.expanduser(root)
        self.transform = transform
        self.split = split
        self.return_spx = return_spx
        self.mask_region = mask_region
        self.dominant_labeling = dominant_labeling
        self.store = store if store is not None else PictureStore()
        self._lut = {}                      # spx path -> ((id(list), len), bool [nseg + 1] on the device)
        self._encode = None                 # id_to_train_id as a device table
        self.get_data_list(datalist, self._load_json(region_dict if region_dict is not None else self.default_region_dict))

    # -- lists ------------------------------------------------------------------------------------
    def get_data_list(self, datalist, json_dict):
        """``im_idx``: [image, label, superpixel] absolute paths per line of the datalist; ``suppix``: superpixel path -> ids of this
        split.  ``None`` (the labelled split starts empty) leaves both empty (:51-76)."""
        self.im_idx, self.suppix = [], {}
        if datalist is not None:
            self.im_idx, self.suppix = formats.read_datalist(datalist, self.root, json_dict, known_ignore=bool(getattr(self.args, 'known_ignore', False)),
                                                             prob_dominant=bool(self.args.prob_dominant))

    def _load_json(self, path):
        return formats.load_region_dict(path)

    # -- label tables -----------------------------------------------------------------------------
    @classmethod
    def encode_target(cls, target):
        """raw label ids -> training ids (host arrays, as the reference's class method :78-81)."""
        return id_to_train_id[np.array(target)]

    @classmethod
    def decode_target(cls, target):
        t = target.clone() if isinstance(target, torch.Tensor) else np.array(target)
        t[target == 255] = 19
        return train_id_to_color[t]

    @classmethod
    def open_spx(cls, spx_fname):
        """int64 [H,W] host array of a superpixel file (png / jpg / pkl)."""
        return formats.open_spx(spx_fname)

    def _encode_on_device(self, raw):
        if self._encode is None or self._encode.device != raw.device:
            self._encode = torch.from_numpy(id_to_train_id_u8).to(raw.device)
        return self._encode[raw.long()].long()

    # -- selection mask ---------------------------------------------------------------------------
    def selection_lut(self, spx_fname, device):
        """bool [nseg + 1] of the ids this split holds for one picture (the pad id ``nseg`` is never set); rebuilt only when the list
        object or its length changed (``expand_training_set`` appends, ``load_datalist`` replaces)."""
        ids = self.suppix.get(spx_fname, [])
        key = (id(ids), len(ids))
        hit = self._lut.get(spx_fname)
        if hit is None or hit[0] != key or hit[1].device != device:
            hit = (key, formats.selection_lut(ids, self.args.nseg, device))
            self._lut[spx_fname] = hit
        return hit[1]

    def selection_mask(self, spx_fname, superpixel):
        lut = self.selection_lut(spx_fname, superpixel.device)
        return lut[superpixel.clamp(min=0, max=self.args.nseg)] & (superpixel >= 0)

    # -- samples ----------------------------------------------------------------------------------
    def sample_files(self, index):
        """The files one sample decodes (the batch provider starts their decodes together): (kind, path) pairs."""
        img, lbl, spx = self.im_idx[index]
        return [('rgb', img), ('map', lbl), ('ids', spx)]

    def prefetch(self, indices):
        self.store.prefetch([f for i in indices for f in self.sample_files(i)])

    def __getitem__(self, index):
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        picture = self.store.picture(img_fname)
        image, (target, superpixel) = self.transform(picture, [self.store.labelmap(lbl_fname), self.store.idmap(spx_fname)])
        target = target.long() if self.dominant_labeling else self._encode_on_device(target)
        if self.mask_region is True:            # keep the labels of the selected superpixels only (:113-127)
            keep = self.selection_mask(spx_fname, superpixel)
            target = torch.where(keep, target, torch.full_like(target, 255))
        sample = {'images': image, 'labels': target, 'fnames': self.im_idx[index]}
        if self.return_spx:
            sample['spx'] = superpixel
        return sample

    def __len__(self):
        return len(self.im_idx)
