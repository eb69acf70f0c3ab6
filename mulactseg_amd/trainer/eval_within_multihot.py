"""Stage-2 evaluation base: labelled-set loader, (num_classes + 1)-way IoU, and the "arg-max within the candidate
set" pseudo label -- reference ``trainer/eval_within_multihot.py:14-146``."""
import numpy as np
import torch

from ..dataloader.utils import DataProvider
from ..models import get_model
from ..utils.miou import MeanIoU
from .base import BaseTrainer


class ActiveTrainer(BaseTrainer):
    predicts_ignore = True
    extra_channels = 1          # the "undefined" channel of the Cityscapes models; 0 in the VOC twin (..._voc.py:21)

    def __init__(self, args, logger, selection_iter):
        self.selection_iter = selection_iter
        super().__init__(args, logger)

    def get_al_model(self):
        a = self.args
        return get_model(model=a.model, num_classes=self.num_classes + self.extra_channels, output_stride=a.output_stride,
                         separable_conv=a.separable_conv, pretrained_backbone=getattr(a, 'pretrained_backbone', True))

    def eval(self, active_set, selection_iter):
        eval_dataset = active_set.trg_label_dataset
        eval_dataset.im_idx = sorted(eval_dataset.im_idx)
        # one process per GPU: every rank generates the pseudo labels of its share of the labelled pictures (round-robin;
        # the PNGs are per-picture files) and the IoU counters are summed over the ranks
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            eval_dataset.im_idx = eval_dataset.im_idx[dist.get_rank()::dist.get_world_size()]
        if getattr(eval_dataset, 'device_resident', False):       # samples are made on the device (file-backed / resident datasets)
            from ..dataloader.utils import ResidentProvider
            self.eval_dataset_loader = ResidentProvider(eval_dataset, batch_size=self.args.val_batch_size, drop_last=False, shuffle=False)
        else:
            self.eval_dataset_loader = DataProvider(dataset=eval_dataset, batch_size=self.args.val_batch_size, shuffle=False,
                                                    num_workers=getattr(self.args, 'val_num_workers', 8), pin_memory=True,
                                                    drop_last=False)
        miou, table = self.inference(loader=self.eval_dataset_loader, prefix='evaluation')
        self.logger.info('[Evaluation Result]')
        self.logger.info('%s' % table)
        self.logger.info('Current eval miou is %.3f %%' % miou)
        return table

    def _batch(self, batch):
        dev = self.device
        return (batch['images'].to(dev, dtype=torch.float32), batch['labels'].to(dev, dtype=torch.long),
                batch['spx'].to(dev), batch['spmask'].to(dev), batch['target'].to(dev))

    def pseudo_labels(self, images, labels, targets, spmasks, superpixels):
        outputs = self.net(images).detach()
        return self.top_pseudo_label_generation(labels, outputs, targets, spmasks, superpixels)

    def inference(self, loader, prefix=''):
        meter = MeanIoU(self.num_classes + 1, self.args.ignore_idx)
        meter._before_epoch()
        self.net.eval()
        with torch.no_grad():
            for _ in range(len(loader)):
                batch = next(loader)
                images, labels, superpixels, spmasks, targets = self._batch(batch)
                plbl = self.pseudo_labels(images, labels, targets, spmasks, superpixels)
                meter._after_step({'outputs': plbl, 'targets': labels})
                self.after_batch(batch, plbl)
        meter.all_reduce(self.device)
        ious = meter._after_epoch()
        miou = np.mean(ious)
        table = ','.join(['%.2f' % miou] + ['%.2f' % v for v in ious])
        print("\n[AL {}-round]: {}\n{}".format(self.selection_iter, prefix, table), flush=True)
        return miou, table

    def after_batch(self, batch, plbl):
        pass

    def top_pseudo_label_generation(self, labels, inputs, targets, spmasks, superpixels):
        """Selected pixels get ``argmax_c (logit_c * Y_c)`` of their superpixel's multi-hot row, others 255
        (``eval_within_multihot.py:93-146``; raw logits, so a row whose target logits are all negative yields the
        first zero entry -- replicated)."""
        N, C, H, W = inputs.shape
        S = targets.shape[1]
        trg = torch.gather(targets.to(inputs.dtype), 1,
                           superpixels.reshape(N, -1, 1).clamp(max=S - 1).expand(N, H * W, C))     # N x HW x C
        scores = inputs.permute(0, 2, 3, 1).reshape(N, -1, C) * trg
        plbl = scores.max(dim=2)[1]
        plbl = torch.where(spmasks.reshape(N, -1), plbl, torch.full_like(plbl, 255))
        return plbl.reshape(N, H, W)
