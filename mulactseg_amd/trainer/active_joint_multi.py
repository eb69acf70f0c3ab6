"""Joint multi-label trainer -- reference ``trainer/active_joint_multi.py:8-76``:
``loss = coeff * merged_positive_CE + group_loss``; the step is skipped when the loss is exactly 0 and
a NaN raises (``check_loss_sanity`` :23-29)."""
import torch

from ..utils.loss import GroupMultiLabelCE, MultiChoiceCE
from . import active


class ActiveTrainer(active.ActiveTrainer):
    def get_criterion(self):
        a = self.args
        self.group_multi_loss = GroupMultiLabelCE(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp)
        self.multi_pos_loss = MultiChoiceCE(num_class=self.num_classes, temperature=a.multi_ce_temp)

    def zero_if_nan(self, loss):
        return 0 if torch.isnan(loss) else loss

    def check_loss_sanity(self, loss):
        """Reference semantics (``active_joint_multi.py:31-37``): a zero loss (no selected pixel in the batch) skips
        the step, NaN raises.  One device->host read instead of the reference's two."""
        v = float(loss.detach())
        if v == 0:
            return False
        if v != v:
            raise ValueError("NaN loss")
        return True

    def update(self, loss):
        if self.check_loss_sanity(loss):
            # DDP averages gradients over ranks; every partial-label loss module already returns the GLOBAL-batch
            # objective (normalisers all-reduced, identical value on every rank -- so this branch is taken by all ranks
            # or by none), hence scale by the world size to get its exact gradient.
            scale = 1
            if self.ddp is not None:
                import torch.distributed as dist
                scale = dist.get_world_size()
            (loss * scale if scale != 1 else loss).backward()
            self.optimizer.step()
        if self.args.scheduler == 'poly':
            self.scheduler.step()

    def update_average_meter(self, values):
        """All meters from ONE device->host transfer (the reference syncs twice per value)."""
        keys = list(values)
        host = torch.stack([values[k].detach().float().reshape(()) for k in keys]).cpu().tolist()
        for key, v in zip(keys, host):
            if v != v:
                raise ValueError("NaN loss")
            if v != 0:
                self.am.add({key: v})

    def _batch(self):
        batch = next(self.train_dataset_loader)
        images = batch['images'].to(self.device, dtype=torch.float32)
        labels = batch['labels'].to(self.device, dtype=self.target_dtype)
        return images, labels, batch['spx'].to(self.device), batch['spmask'].to(self.device)

    def train_impl(self, total_itrs, val_period):
        self.net.train()
        for iteration in range(total_itrs):
            images, labels, superpixels, spmasks = self._batch()
            self.optimizer.zero_grad()
            preds = self.forward_train(images)
            group_loss = self.group_multi_loss(preds, labels, superpixels, spmasks)
            pos_loss = self.multi_pos_loss(preds, labels, superpixels, spmasks)
            loss = self.args.coeff * pos_loss + group_loss
            self.update(loss)
            self.update_average_meter({'train-loss': loss, 'pos-loss': pos_loss, 'group-loss': group_loss})
            self.log_training(iteration, None, total_itrs)
            self.log_validation(iteration, val_period)
