"""Stage-2 pseudo labels for PASCAL VOC -- reference ``trainer/eval_save_cosplbl_prop_includeonehot_voc.py``: the
Cityscapes generator on the VOC evaluation base (21 channels, no "undefined" channel); the saved PNG goes through a
NEAREST resize to ``batch['imsizes']``, which the reference asserts to be its current size (:77-81)."""
from . import eval_save_cosplbl_prop_includeonehot, eval_within_multihot_voc


class ActiveTrainer(eval_save_cosplbl_prop_includeonehot.ActiveTrainer, eval_within_multihot_voc.ActiveTrainer):
    extra_channels = 0

    def after_batch(self, batch, plbl):
        from PIL import Image
        fname = batch['fnames'][0][1]
        lbl_id = fname.split('/')[-1].split('.')[0]
        im = Image.fromarray(plbl[0].cpu().numpy().astype('uint8'))
        if 'imsizes' in batch:
            w, h = [int(v) for v in batch['imsizes'][0]]              # (:77) the loader records (width, height)
            assert [w, h] == list(im.size)                             # (:79) -- so the resize below is the identity
            im = im.resize((w, h), Image.NEAREST)
        im.save("{}/{}.png".format(self._save_dir(), lbl_id))
