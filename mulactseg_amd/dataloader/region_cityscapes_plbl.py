"""Stage-2 training set: pictures + the pseudo-label PNGs the stage-2 generator wrote -- the reference's
``dataloader/region_cityscapes_plbl.py:18-47``.  The PNG directory is derived from ``--resume_checkpoint``
(``<dir>/plbl_gen[_<plbl_type>]/round_RR``, RR = the two digits before the extension), the file name from the picture's
(``<city>_<seq>_<frame>.png``); labels are used as stored (training ids incl. the "undefined" class 19, 255 = no label)."""
import os

from . import region_cityscapes


def plbl_root_of(args):
    rnd = args.resume_checkpoint[-6:-4]
    assert int(rnd) == args.init_iteration
    base = '/'.join(args.resume_checkpoint.split('/')[:-1])
    ptype = getattr(args, 'plbl_type', None)
    return '{}/plbl_gen{}/round_{}'.format(base, '' if ptype is None else '_' + ptype, rnd)


class RegionCityscapes(region_cityscapes.RegionCityscapes):
    def __init__(self, args, root, datalist, split='train', transform=None, return_spx=False,
                 region_dict=None, mask_region=True, dominant_labeling=False, store=None):
        super().__init__(args, root, datalist, split, transform, return_spx, region_dict, mask_region, dominant_labeling, store=store)
        self.plbl_root = plbl_root_of(args)
        assert os.path.exists(self.plbl_root), "no pseudo labels at %s (run the stage-2 generator first)" % self.plbl_root

    def plbl_file(self, img_fname):
        return "{}/{}.png".format(self.plbl_root, img_fname.split('/')[-1].split('_leftImg8bit')[0])

    def sample_files(self, index):
        img = self.im_idx[index][0]
        return [('rgb', img), ('map', self.plbl_file(img))]

    def __getitem__(self, index):
        img_fname = self.im_idx[index][0]
        image, (target,) = self.transform(self.store.picture(img_fname), [self.store.labelmap(self.plbl_file(img_fname))])
        return {'images': image, 'labels': target.long(), 'fnames': self.im_idx[index]}
