"""Region dataset whose pictures and superpixel maps are resident in HBM -- the device-side counterpart of the
reference's ``dataloader/region_cityscapes_or_tensor.py:15-100`` (``RegionCityscapesOrTensor``).

The reference opens a PNG and a superpixel file per ``__getitem__`` and augments PIL images in DataLoader workers.
With 288 GB of HBM the decoded training set fits once (Cityscapes: 2 975 pictures x 6 MB u8 + 4 MB u16 maps = 30 GB),
so a sample is produced by one kernel (``DeviceTrainAugment``) plus a table lookup for the selection mask; no worker
processes, no host->device copy of float crops.  The sample dictionary and the ``im_idx`` / ``suppix`` bookkeeping are
the reference's, so ``RegionActiveDataset`` and the trainers use it unchanged."""
import numpy as np
import torch

from .device_transforms import DeviceTrainAugment
from .formats import selection_lut


class ResidentRegionDataset(torch.utils.data.Dataset):
    device_resident = True          # trainers batch it with ResidentProvider instead of a DataLoader with workers
    def __init__(self, args, pictures, superpixels, multi_hot_cls, names, split='active-label', region_dict=None, rng=None):
        """pictures: list of uint8 [H,W,3] device tensors; superpixels: list of integer [H,W] device tensors;
        multi_hot_cls: uint8 [n_img, nseg, num_classes + 1]; names: list of (img, lbl, spx) path strings (the keys of
        ``suppix``); region_dict: ``{spx path: list of ids}`` of the regions this split starts with."""
        assert split in ('active-label', 'active-ulabel')
        assert len(pictures) == len(superpixels) == len(names)
        self.args = args
        self.split = split
        self.mask_region = True
        self.pictures, self.superpixels = list(pictures), list(superpixels)
        self.multi_hot_cls = multi_hot_cls
        self.names = {n[2]: k for k, n in enumerate(names)}
        # rows of multi_hot_cls by label-file stem, as RegionActiveDataset looks them up (region_cityscapes_or_tensor.py:41-46)
        self.id_to_index = {n[2].split('/')[-1].split('.')[0]: k for k, n in enumerate(names)}
        # entries are LISTS like the reference's (region_cityscapes.py:75): expand_training_set compares them with
        # ``joined.split(',')``, and a list never equals a tuple
        self.im_idx = [list(n) for n in names] if split == 'active-ulabel' else []
        self.suppix = {}
        for n in names:
            ids = list(region_dict.get(n[2], [])) if region_dict is not None else []
            if split == 'active-ulabel' and region_dict is None:
                ids = list(range(args.nseg))
            if ids:
                self.suppix[n[2]] = ids
                if split == 'active-label':
                    self.im_idx.append(list(n))
        if split == 'active-ulabel':
            self.isselected = np.zeros((len(names), args.nseg), dtype=np.uint8)     # region_active_dataset.py:55-56
        self._lut = {}              # spx path -> (number of ids it was built from, bool [nseg + 1] on the device)
        self.transform = DeviceTrainAugment(size=(768, 768), scale_range=(0.5, 2.0), pad_values=[args.nseg], rng=rng)
        self.pool_transform = DeviceTrainAugment(scale_range=(1.0, 1.0), pad_values=[args.nseg])

    def __len__(self):
        return len(self.im_idx)

    def _slot(self, spx_fname):
        return self.names[spx_fname]

    def _selection_lut(self, spx_fname, device):
        """bool [nseg + 1] table of the selected ids of one picture, kept on the device and rebuilt only when the id list
        grew (expand_training_set appends; a pad id ``nseg`` is never selected) -- a sample then costs no H2D copy."""
        ids = self.suppix.get(spx_fname, [])
        hit = self._lut.get(spx_fname)
        key = (id(ids), len(ids))           # load_datalist replaces the lists wholesale: a new list object is a new table
        if hit is None or hit[0] != key:
            hit = (key, selection_lut(ids, self.args.nseg, device))
            self._lut[spx_fname] = hit
        return hit[1]

    def __getpoolitem__(self, k):
        """Normalised full-size picture + untouched map (``region_cityscapes_or_tensor.py:47-52``)."""
        pic, spx = self.pictures[k], self.superpixels[k]
        H, W = pic.shape[:2]
        t = self.pool_transform
        t.size = (H, W)
        params = dict(scale=1.0, th=H, tw=W, gap_y=0, gap_x=0, i=0, j=0, flip=False)
        image, (s,) = t(pic, [spx], params=params)
        return {'images': image, 'spx': s, 'labels': self.multi_hot_cls[k]}

    def __getitem__(self, index):
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        k = self._slot(spx_fname)
        if self.split == 'active-ulabel':
            return self.__getpoolitem__(k)
        image, (superpixel,) = self.transform(self.pictures[k], [self.superpixels[k]])
        sp_mask = self._selection_lut(spx_fname, superpixel.device)[superpixel.clamp(min=0, max=self.args.nseg)] & (superpixel >= 0)  # (:88-89)
        return {'images': image, 'labels': self.multi_hot_cls[k], 'spx': superpixel, 'spmask': sp_mask,
                'fnames': self.im_idx[index]}
