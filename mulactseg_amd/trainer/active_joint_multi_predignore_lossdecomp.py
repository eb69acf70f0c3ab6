"""The production stage-1 trainer (Cityscapes) -- reference
``trainer/active_joint_multi_predignore_lossdecomp.py:74-116``:

    loss = coeff * ce(one-hot regions) + coeff_mc * mc(multi-hot regions) + coeff_gm * group(multi-hot regions)

The three losses come from ONE forward scan and ONE backward scan of the logits
(``utils.loss.FusedPartialLabelLoss`` over ``csrc/losses.hip``) instead of two modules that each
re-compute the softmax; by default (``args.lowres_loss``, True) the scans take the model's quarter-resolution logits and
evaluate the final x4 bilinear upsampling (``models/segmentation/utils.py:25``) per selected pixel.  Under data parallelism the normalisers ``1 + n`` are global over the batch
(SURVEY.md section 5.8): the fixed-point sums and counts are all-reduced before the division so that
N GPUs x batch 4 optimise exactly the single-GPU objective of batch 4N.
"""

from ..utils.loss import FusedPartialLabelLoss, GroupMultiLabelCE_onlymulti, OnehotCEMultihotChoice
from . import active_joint_multi_predignore


class ActiveTrainer(active_joint_multi_predignore.ActiveTrainer):
    def get_criterion(self):
        a = self.args
        # the reference's two modules stay available under their names (drop-in surface) ...
        self.group_multi_loss = GroupMultiLabelCE_onlymulti(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp)
        self.multi_pos_loss = OnehotCEMultihotChoice(num_class=self.num_classes, temperature=a.multi_ce_temp)
        # ... and train_impl uses the fused scan when both temperatures agree (they do in every script)
        self.fused_loss = None
        if a.group_ce_temp == a.multi_ce_temp:
            self.fused_loss = FusedPartialLabelLoss(a.nseg, a.group_ce_temp, a.multi_ce_temp, only_multi=True, decomp=True)

    def losses(self, preds, labels, superpixels, spmasks):
        if self.fused_loss is not None:
            return self.fused_loss(preds, labels, superpixels, spmasks)
        group = self.group_multi_loss(preds, labels, superpixels, spmasks)
        ce, mc = self.multi_pos_loss(preds, labels, superpixels, spmasks)
        return group, ce, mc

    def train_impl(self, total_itrs, val_period):
        a = self.args
        self.net.train()
        for iteration in range(total_itrs):
            images, labels, superpixels, spmasks = self._batch()
            self.optimizer.zero_grad()
            if self.fused_loss is not None and getattr(a, 'lowres_loss', True) and images.is_cuda:
                # quarter-resolution logits in, upsampling inside the loss scans (no [N,C,H,W] logit / gradient tensors)
                preds_q = self.forward_train(images, lowres=True)
                loss, group_loss, ce_loss, mc_loss = self.fused_loss.weighted_lowres(preds_q, images.shape[-2:], labels, superpixels, spmasks,
                                                                                     a.coeff, a.coeff_mc, a.coeff_gm)
            else:
                preds = self.forward_train(images)
                group_loss, ce_loss, mc_loss = self.losses(preds, labels, superpixels, spmasks)
                loss = (a.coeff * ce_loss) + (a.coeff_mc * mc_loss) + (a.coeff_gm * group_loss)
            self.update(loss)
            self.update_average_meter({'train-loss': loss, 'ce-loss': ce_loss, 'pos-loss': mc_loss, 'group-loss': group_loss})
            self.log_training(iteration, None, total_itrs)
            self.log_validation(iteration, val_period)
        self.flush_meters()
        self.check_stream_k()
