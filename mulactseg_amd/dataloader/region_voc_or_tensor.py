"""VOC with the precomputed multi-hot superpixel labels -- the reference's ``dataloader/region_voc_or_tensor.py:16-111``
(``RegionVOCOr``; ``--loader region_voc_or_tensor --train_transform rescale_513_multi_notrg``).  Differences from the Cityscapes
twin: ``--nseg 150 / 600`` name the SEEDS directories ``seeds_32 / seeds_16`` (:30-35), the last ("undefined") column of the tensor
is dropped because the VOC models do not predict it (:50-53), and the rows are indexed by the bare picture names of
``args.trg_datalist`` (:55-62).  (The reference's untrimmed path swaps its two format arguments, :42-43 -- the layout below is the
one its trimmed path and its generator scripts use.)"""
from . import region_cityscapes_or_tensor, region_voc

_SEEDS_DIR = {150: 32, 600: 16}


class RegionVOCOr(region_cityscapes_or_tensor.RegionCityscapesOr, region_voc.RegionVOC):
    default_region_dict = region_voc.RegionVOC.default_region_dict

    def multi_hot_files(self):
        a = self.args
        if a.nseg not in _SEEDS_DIR:
            raise NotImplementedError("VOC superpixels exist for --nseg 150 and 600")
        name = "gtFine_multi_tensor_trim_{0}x{0}".format(a.trim_kernel_size) if getattr(a, 'trim_multihot_boundary', False) else "multihot"
        base = '{}/superpixels/pascal_voc_seg/seeds_{}/train/{}'.format(self.root, _SEEDS_DIR[a.nseg], name)
        return base + '/multi_hot_cls.npy', base + '/sp_size.npy'

    def prepare_multi_hot(self, table):
        return table[:, :, :-1].contiguous()

    def label_rows(self):
        with open(self.args.trg_datalist, 'r') as f:
            return {line.split('/')[-1].split('.')[0]: i for i, line in enumerate(l for l in f.read().splitlines() if l)}
