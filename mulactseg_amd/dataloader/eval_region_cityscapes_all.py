"""The labelled set as the stage-2 pseudo-label generator reads it -- the reference's
``dataloader/eval_region_cityscapes_all.py:11-68`` (``--loader eval_region_cityscapes_all --train_transform eval_spx``): per picture
the precise ground truth with "ignore" turned into class 19 (``labels``, only for the IoU of the generated labels), the multi-hot
table (``target``), the id map and the mask of the selected superpixels -- without the one-hot ones unless the trainer saves labels
(``'eval_save' in args.method``)."""
import torch

from . import region_cityscapes_or_tensor


class RegionCityscapesOr(region_cityscapes_or_tensor.RegionCityscapesOr):
    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, loading='binary', load_smaller_spx=False, store=None):
        super().__init__(args, root, datalist, split, transform, return_spx, region_dict, mask_region, dominant_labeling, loading,
                         load_smaller_spx, store=store)
        assert self.mask_region
        self.remove_dominant = 'eval_save' not in args.method

    def precise_label_file(self, lbl_fname):
        stem = lbl_fname.split('/')[-1].split('.')[0]
        return '{}/gtFine/train/{}/{}_gtFine_labelIds.png'.format(self.root, stem.split('_')[0], stem)

    def sample_files(self, index):
        img, lbl, spx = self.im_idx[index]
        return [('rgb', img), ('map', self.precise_label_file(lbl)), ('ids', spx)]

    def __getitem__(self, index):
        img_fname, lbl_fname, spx_fname = self.im_idx[index]
        picture = self.store.picture(img_fname)
        raw = self.store.labelmap(self.precise_label_file(lbl_fname))
        image, (precise, superpixel) = self.transform(picture, [raw, self.store.idmap(spx_fname)])
        precise = self._encode_on_device(precise)
        precise = torch.where(precise == 255, torch.full_like(precise, 19), precise)        # "undefined" is a class here (:37-41)
        target = self.multi_hot_row(lbl_fname, image.device)
        keep = self.selection_lut(spx_fname, image.device)
        if self.remove_dominant:                                    # drop the selected superpixels with exactly one class (:55-58)
            keep = keep.clone()
            keep[:-1] &= target.sum(dim=1) != 1
        sp_mask = keep[superpixel.clamp(min=0, max=self.args.nseg)] & (superpixel >= 0)
        return {'images': image, 'labels': precise, 'target': target, 'spx': superpixel, 'spmask': sp_mask, 'fnames': self.im_idx[index]}
