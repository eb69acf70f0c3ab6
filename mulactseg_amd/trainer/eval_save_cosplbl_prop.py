"""Stage-2 pseudo-label generation: cosine prototypes + one-ring propagation over the multi-hot superpixels, saved
as uint8 PNGs -- reference ``trainer/eval_save_cosplbl_prop.py:20-314``.

The reference materialises ``feat_forward``'s 256-channel full-resolution features (2.1 GB per Cityscapes image) and
walks the selected superpixels in Python with a CPU dilation each; here the quarter-resolution features go straight
into the K9 kernels (``ops.stage2_pseudo_labels``, ``csrc/stage2.hip``)."""
import os


from .. import ops
from . import eval_within_multihot


class ActiveTrainer(eval_within_multihot.ActiveTrainer):
    include_onehot = False

    def __init__(self, args, logger, selection_iter):
        super().__init__(args, logger, selection_iter)
        assert args.val_batch_size == 1
        self.save_dir = None

    def _save_dir(self):
        """``<dir of init_checkpoint>/plbl_gen[_<plbl_type>]/round_RR`` (:33-39)."""
        if self.save_dir is None:
            ckpt = self.args.init_checkpoint
            rnd = ckpt.split('/')[-1][-6:-4]
            base = '/'.join(ckpt.split('/')[:-1])
            ptype = getattr(self.args, 'plbl_type', None)
            self.save_dir = '{}/plbl_gen{}/round_{}'.format(base, '' if ptype is None else '_' + ptype, rnd)
            os.makedirs(self.save_dir, exist_ok=True)
        return self.save_dir

    def pseudo_labels(self, images, labels, targets, spmasks, superpixels):
        feats, outputs = self.net.feat_forward_lowres(images)
        return self.pseudo_label_generation(labels, feats, outputs, targets, spmasks, superpixels)

    def pseudo_label_generation(self, labels, feats, inputs, targets, spmasks, superpixels):
        """Same signature as the reference (:121); ``feats`` may be the quarter-resolution map."""
        return ops.stage2_pseudo_labels(feats.contiguous(), inputs.contiguous(), targets.contiguous(), spmasks.contiguous(),
                                        superpixels.contiguous(), include_onehot=self.include_onehot)

    def after_batch(self, batch, plbl):
        from PIL import Image
        fname = batch['fnames'][0][1]
        lbl_id = fname.split('/')[-1].split('.')[0]
        Image.fromarray(plbl[0].cpu().numpy().astype('uint8')).save("{}/{}.png".format(self._save_dir(), lbl_id))
