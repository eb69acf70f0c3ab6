"""PASCAL VOC trainer -- reference ``trainer/active_voc.py``: ``trainer/active.py`` on ``base_voc`` plus the
``--freeze_bn`` option (:75-76: BatchNorm layers in eval mode with frozen affine parameters during training; a frozen
BatchNorm inside the training graph runs on the PyTorch ops, ``ops.bn_act_supported``)."""
from ..models import freeze_bn
from . import active


class ActiveTrainer(active.ActiveTrainer):
    def train_impl(self, total_itrs, val_period):
        if getattr(self.args, 'freeze_bn', False) is True:
            _train = self.net.train

            def train_keep_bn_frozen(mode=True):          # log_validation() calls net.train() again after each eval
                out = _train(mode)
                if mode:
                    freeze_bn(self.net)
                return out
            self.net.train = train_keep_bn_frozen
        return super().train_impl(total_itrs, val_period)
