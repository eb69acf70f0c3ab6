"""IoU meters with the reference's interface (``utils/miou.py``, ``utils/miou_evalignore.py``), counting
on the device (``csrc/metrics.hip``) instead of 19 x 3 host-synchronising reductions per batch.

``MeanIoU(num_classes, ignore_label)``: ``_before_epoch()``, ``_after_step({'outputs', 'targets'})``,
``_after_epoch()`` -> per-class IoU x 100 where an unseen class counts as 100 (``utils/miou.py:63-70``).
``IoUIgnore``: IoU of the extra "undefined" class (label ``num_classes`` in the prediction,
``ignore_label`` in the ground truth).
``LogitsIoU`` (new): both meters from one read of the logits (fused arg-max).
The only host synchronisation is in ``_after_epoch`` / ``total_*`` (one copy of 3C+3 integers).
"""
import numpy as np
import torch

from .. import ops


class MeanIoU:
    def __init__(self, num_classes, ignore_label, output_tensor='outputs', target_tensor='targets', name='iou'):
        self.num_classes = num_classes
        self.ignore_label = ignore_label
        self.name = name
        self.output_tensor = output_tensor
        self.target_tensor = target_tensor
        self._counts = None

    def _before_epoch(self):
        self._counts = None

    def _ensure(self, device):
        if self._counts is None:
            self._counts = torch.zeros(3 * self.num_classes + 3, dtype=torch.int64, device=device)
        return self._counts

    def _after_step(self, output_dict):
        outputs = output_dict[self.output_tensor]
        targets = output_dict[self.target_tensor]
        ops.iou_counts(outputs.contiguous(), None, targets.contiguous(), self.num_classes, self.ignore_label,
                       self._ensure(targets.device))

    def _host_counts(self):
        if self._counts is None:
            return np.zeros(3 * self.num_classes + 3, dtype=np.float64)
        return self._counts.cpu().numpy().astype(np.float64)

    def all_reduce(self, device=None):
        """Sum the integer counters over the ranks of an initialised process group (validation / stage-2 sets are
        sharded by rank).  A rank that saw no batch contributes zeros (``device`` tells it where to allocate them)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            if self._counts is None and device is not None:
                self._ensure(device)
            if self._counts is not None:
                dist.all_reduce(self._counts)

    @property
    def total_seen(self):
        return self._host_counts()[:self.num_classes]

    @property
    def total_correct(self):
        return self._host_counts()[self.num_classes:2 * self.num_classes]

    @property
    def total_positive(self):
        return self._host_counts()[2 * self.num_classes:3 * self.num_classes]

    def _after_epoch(self, ignore_label_list=None):
        c = self._host_counts()
        C = self.num_classes
        ious = []
        for i in range(C):
            if ignore_label_list is not None and i in ignore_label_list:
                continue
            seen, correct, positive = c[i], c[C + i], c[2 * C + i]
            ious.append(1 if seen == 0 else correct / (seen + positive - correct))
        return [v * 100 for v in ious]

    def _after_epoch_ipr(self):
        c = self._host_counts()
        C = self.num_classes
        ious, precs, recs = [], [], []
        for i in range(C):
            seen, correct, positive = c[i], c[C + i], c[2 * C + i]
            if seen == 0:
                ious.append(1); precs.append(1); recs.append(1)
            else:
                ious.append(correct / (seen + positive - correct))
                precs.append(correct / positive)
                recs.append(correct / seen)
        return ([v * 100 for v in ious], [v * 100 for v in precs], [v * 100 for v in recs])


class IoUIgnore(MeanIoU):
    """IoU of the "undefined" class -- ``utils/miou_evalignore.py:8-62``."""

    def _after_step(self, output_dict):
        outputs_all = output_dict[self.output_tensor]
        targets_all = output_dict[self.target_tensor]
        assert type(outputs_all) == torch.Tensor
        ops.iou_counts(None, outputs_all.contiguous(), targets_all.contiguous(), self.num_classes, self.ignore_label,
                       self._ensure(targets_all.device))

    def _ignore_counts(self):
        c = self._host_counts()
        return c[3 * self.num_classes], c[3 * self.num_classes + 1], c[3 * self.num_classes + 2]

    @property
    def total_seen(self):
        return self._ignore_counts()[0]

    @property
    def total_correct(self):
        return self._ignore_counts()[1]

    @property
    def total_positive(self):
        return self._ignore_counts()[2]

    def _after_epoch(self, ignore_label_list=None):
        seen, correct, positive = self._ignore_counts()
        if seen == 0:
            return 100.0
        return correct / (seen + positive - correct) * 100


class LogitsIoU(MeanIoU):
    """Both meters of ``ActiveTrainer.inference`` (``trainer/active_joint_multi_predignore.py:175-215``)
    from one pass over the logits: ``step(logits, labels)``; ``ious()`` / ``ignore_iou()``."""

    def step(self, logits, labels):
        ops.logits_iou_counts(logits.contiguous(), labels.contiguous(), self.num_classes, self.ignore_label,
                              self._ensure(logits.device))

    def ious(self):
        return self._after_epoch()

    def ignore_iou(self):
        c = self._host_counts()
        seen, correct, positive = c[3 * self.num_classes:3 * self.num_classes + 3]
        return 100.0 if seen == 0 else correct / (seen + positive - correct) * 100
