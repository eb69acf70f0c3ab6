"""Evaluation with sliding-window inference (crop 800, stride 2/3) -- reference ``trainer/eval_slide.py:17-88``."""
import numpy as np
import torch

from ..utils.miou import LogitsIoU
from ..utils.sliding_evaluator import SlidingEval
from . import active


class ActiveTrainer(active.ActiveTrainer):
    crop_size = 800
    stride_rate = 2 / 3

    def inference(self, loader, prefix=''):
        """mIoU of arg-max over the summed window scores, one image at a time (:55-88)."""
        helper = LogitsIoU(self.num_classes, self.args.ignore_idx)
        helper._before_epoch()
        self.net.eval()
        evaluator = SlidingEval(model=self.net, crop_size=self.crop_size, stride_rate=self.stride_rate, device=self.device,
                                class_number=self.num_classes)
        with torch.no_grad():
            for _ in range(len(loader)):
                batch = next(loader)
                images = batch['images'].to(self.device, dtype=torch.float32)
                labels = batch['labels'].to(self.device, dtype=torch.long)
                for i in range(images.shape[0]):
                    scores = evaluator(images[i:i + 1])
                    helper.step(scores[None].contiguous(), labels[i:i + 1])
        helper.all_reduce(self.device)
        ious = helper.ious()
        miou = float(np.mean(ious))
        table = ','.join(['%.2f' % miou] + ['%.2f' % v for v in ious])
        print("\n[AL {}-round]: {}\n{}".format(self.selection_iter, prefix, table), flush=True)
        return miou, table
