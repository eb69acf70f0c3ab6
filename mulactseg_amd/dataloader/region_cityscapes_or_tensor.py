"""File-backed Cityscapes dataset with the precomputed multi-hot superpixel labels -- the reference's
``dataloader/region_cityscapes_or_tensor.py:16-96`` (``RegionCityscapesOr``): the stage-1 production loader
(``--loader region_cityscapes_or_tensor --train_transform rescale_769_multi_notrg``).

``multi_hot_cls.npy`` (u8 ``[n_img, nseg, num_classes + 1]``, rows in ``args.trg_datalist`` order) is read once; a labelled sample is
the augmented picture, the augmented id map, the picture's whole multi-hot table and the mask of selected superpixels; a pool sample
is the resized picture + id map.  All tensors of a sample are on the device (see ``region_cityscapes.py``)."""
import numpy as np
import torch

from . import formats
from .region_cityscapes import RegionCityscapes


class RegionCityscapesOr(RegionCityscapes):
    #: directory pattern of the label tensors under the data root (:31-36)
    tensor_dir = '{root}/superpixel_seed/cityscapes/{spx_method}_{nseg}/train/{name}'

    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, loading='binary', load_smaller_spx=False, store=None):
        super().__init__(args, root, datalist, split, transform, return_spx, region_dict, mask_region, dominant_labeling, store=store)
        self.loading = loading
        if load_smaller_spx:
            raise NotImplementedError("--load_smaller_spx (a second, finer superpixel map per sample) is outside the hot path")
        self.load_smaller_spx = False
        assert not (getattr(args, 'ignore_size', 0) != 0 and getattr(args, 'mark_topk', -1) != -1)
        mh_path, _ = self.multi_hot_files()
        self.multi_hot_cls = self.prepare_multi_hot(torch.from_numpy(np.load(mh_path)))      # (n_img, nseg, n_cls [+ 1])
        self.isselected = np.zeros(tuple(self.multi_hot_cls.shape[:-1]), dtype=np.uint8)     # (n_img, nseg), region_active_dataset.py:55-56
        self.id_to_index = self.label_rows()
        self._mh_dev = None

    def multi_hot_files(self):
        a = self.args
        name = "gtFine_multi_tensor"
        if getattr(a, 'trim_multihot_boundary', False):
            name += "_trim_{0}x{0}".format(a.trim_kernel_size)
        base = self.tensor_dir.format(root=self.root, spx_method=getattr(a, 'spx_method', 'seeds'), nseg=a.nseg, name=name)
        return base + '/multi_hot_cls.npy', base + '/sp_size.npy'

    def prepare_multi_hot(self, table):
        return table

    def label_rows(self):
        """label-file stem -> row of the tensor, from the full target datalist (:41-46)."""
        return formats.id_to_index(self.args.trg_datalist)

    def multi_hot_row(self, lbl_fname, device):
        if self._mh_dev is None or self._mh_dev.device != device:
            self._mh_dev = self.multi_hot_cls.to(device)            # 116 MB for Cityscapes: resident once
        return self._mh_dev[self.id_to_index[lbl_fname.split('/')[-1].split('.')[0]]]

    def sample_files(self, index):
        img, _, spx = self.im_idx[index]
        return [('rgb', img), ('ids', spx)]

    def __getitem__(self, index):
        assert self.mask_region
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        picture = self.store.picture(img_fname)
        image, (superpixel,) = self.transform(picture, [self.store.idmap(spx_fname)])
        target = self.multi_hot_row(lbl_fname, image.device)
        if self.split == 'active-ulabel':                           # pool sample (:47-52,72-73)
            return {'images': image, 'spx': superpixel, 'labels': target}
        return {'images': image, 'labels': target, 'spx': superpixel, 'spmask': self.selection_mask(spx_fname, superpixel),
                'fnames': self.im_idx[index]}
