"""VOC stage-2 training set (pictures + generated pseudo-label PNGs) -- the reference's ``dataloader/region_voc_plbl.py:18-47``; the
PNG is named after the picture (``<name>.png``)."""
import os

from . import region_voc
from .region_cityscapes_plbl import plbl_root_of


class RegionVOC(region_voc.RegionVOC):
    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, store=None):
        super().__init__(args, root, datalist, split, transform, return_spx, region_dict, mask_region, dominant_labeling, store=store)
        self.plbl_root = plbl_root_of(args)
        assert os.path.exists(self.plbl_root), "no pseudo labels at %s (run the stage-2 generator first)" % self.plbl_root

    def plbl_file(self, img_fname):
        return "{}/{}.png".format(self.plbl_root, img_fname.split('/')[-1].split('.')[0])

    def sample_files(self, index):
        img = self.im_idx[index][0]
        return [('rgb', img), ('map', self.plbl_file(img))]

    def __getitem__(self, index):
        img_fname = self.im_idx[index][0]
        image, (target,) = self.transform(self.store.picture(img_fname), [self.store.labelmap(self.plbl_file(img_fname))])
        return {'images': image, 'labels': target.long(), 'fnames': self.im_idx[index]}
