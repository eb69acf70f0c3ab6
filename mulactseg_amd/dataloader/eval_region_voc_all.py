"""The VOC labelled set as the stage-2 generator reads it -- the reference's ``dataloader/eval_region_voc_all.py:11-78``: the class
PNG with "ignore" (255) turned into class 21, the multi-hot table, the id map, the mask of the selected superpixels that carry at
least one class (minus the one-hot ones unless labels are being saved) and the picture's original (w, h)."""
import os

import torch

from . import region_voc, region_voc_or_tensor


class RegionVOCOr(region_voc_or_tensor.RegionVOCOr):
    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, loading='binary', load_smaller_spx=False, store=None):
        super().__init__(args, root, datalist, split, transform, return_spx, region_dict, mask_region, dominant_labeling, loading,
                         load_smaller_spx, store=store)
        assert self.mask_region
        self.remove_dominant = 'eval_save' not in args.method

    def precise_label_file(self, lbl_fname):
        name = lbl_fname.split('/')[-1].split('.')[0]
        return region_voc.voc_paths(self.root, name, self.dominant_labeling)[1]

    def sample_files(self, index):
        img, lbl, spx = self.im_idx[index]
        return [('rgb', img), ('map', self.precise_label_file(lbl)), ('ids', spx)]

    def __getitem__(self, index):
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        picture = self.store.picture(img_fname)
        raw = self.store.labelmap(self.precise_label_file(lbl_fname))
        image, (precise, superpixel) = self.transform(picture, [raw, self.store.idmap(spx_fname)])
        precise = precise.long()
        precise = torch.where(precise == 255, torch.full_like(precise, 21), precise)
        target = self.multi_hot_row(lbl_fname, image.device)
        n_cls = target.sum(dim=1)
        keep = self.selection_lut(spx_fname, image.device).clone()
        keep[:-1] &= n_cls != 0                                     # (:63-64)
        if self.remove_dominant:
            keep[:-1] &= n_cls != 1
        sp_mask = keep[superpixel.clamp(min=0, max=self.args.nseg)] & (superpixel >= 0)
        return {'images': image, 'labels': precise, 'target': target, 'spx': superpixel, 'spmask': sp_mask,
                'imsizes': (int(picture.shape[1]), int(picture.shape[0])), 'fnames': self.im_idx[index]}
