"""File-backed PASCAL VOC 2012 region dataset -- the reference's ``dataloader/region_voc.py:34-176`` (``RegionVOC``).  The datalist
holds bare picture names; the three paths are composed from the data root (:75-84): ``VOC2012/JPEGImages/<name>.jpg``, the class
PNG (``VOC2012/SegmentationClass`` or the dominant-label PNG of the superpixel directory) and the SEEDS superpixel pickle; the region
dictionary is keyed by the bare name.  Labels are class indices already (no encoding).  Samples are made on the device exactly as
in ``region_cityscapes.py``."""
import os

import numpy as np
import torch

from . import formats, region_cityscapes
from .constant import voc_id_to_color_map

SPX_DIR = 'superpixels/pascal_voc_seg/seeds_32/train'


def voc_paths(root, name, dominant_labeling):
    lbl = os.path.join(root, SPX_DIR, 'gtFine_dominant', name + '.png') if dominant_labeling else \
        os.path.join(root, 'VOC2012/SegmentationClass', name + '.png')
    return [os.path.join(root, 'VOC2012/JPEGImages', name + '.jpg'), lbl, os.path.join(root, SPX_DIR, 'label', name + '.pkl')]


class RegionVOC(region_cityscapes.RegionCityscapes):
    default_region_dict = "dataloader/init_data/voc/train_seed32.dict"

    def get_data_list(self, datalist, json_dict):
        self.im_idx, self.suppix = [], {}
        if datalist is None:
            return
        with open(datalist, 'r') as f:
            names = [line.split('\t')[0] for line in f.read().splitlines() if line]
        for name in names:
            paths = voc_paths(self.root, name, self.dominant_labeling)
            self.im_idx.append(paths)
            self.suppix[paths[2]] = json_dict[name]

    @classmethod
    def encode_target(cls, target):
        return np.array(target)

    @classmethod
    def decode_target(cls, target):
        t = target.clone() if isinstance(target, torch.Tensor) else np.array(target)
        t[target == 255] = 21
        return voc_id_to_color_map[t]

    def _encode_on_device(self, raw):
        return raw.long()
