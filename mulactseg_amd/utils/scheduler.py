"""Polynomial learning-rate decay -- reference ``utils/scheduler.py:4-13``:
``lr = max(base_lr * (1 - it/max_iters) ** power, min_lr)``."""
from torch.optim.lr_scheduler import _LRScheduler


class PolyLR(_LRScheduler):
    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power = power
        self.max_iters = max_iters
        self.min_lr = min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        decay = (1 - self.last_epoch / self.max_iters) ** self.power
        return [max(base_lr * decay, self.min_lr) for base_lr in self.base_lrs]
