"""Flags, seeding and running averages the hot-path plugins read -- the in-scope subset of the
reference's ``utils/common.py`` (``get_parser`` :208-370 carries ~110 flags, most of them for ablation
trainers that are out of scope; every flag below keeps the reference's name and default)."""
import argparse
import os
import random

import numpy as np
import torch


class AverageMeter:
    """key -> running mean, with pop-and-reset -- ``utils/common.py:21-57``."""

    def __init__(self, *keys):
        self._data = {k: [0.0, 0] for k in keys}

    def add(self, values, denominator=None):
        for k, v in values.items():
            slot = self._data.setdefault(k, [0.0, 0])
            slot[0] += v
            slot[1] += 1 if denominator is None else denominator

    def get(self, *keys):
        if len(keys) == 1:
            total, n = self._data.get(keys[0], (0.0, 0))
            return total / n if n else 0
        return tuple(self.get(k) for k in keys)

    def pop(self, key=None):
        if key is None:
            for k in self._data:
                self._data[k] = [0.0, 0]
            return None
        v = self.get(key)
        self._data[key] = [0.0, 0]
        return v

    def get_whole_data(self):
        return self._data


def seed_everything(seed):
    """``utils/common.py:59-67``."""
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def get_parser():
    p = argparse.ArgumentParser(description="MulActSeg hot path on MI355X")
    a = p.add_argument
    # model / plugins
    a("-m", "--model", type=str, default='deeplabv3plus_resnet50')
    a("--separable_conv", action='store_true', default=False)
    a("--output_stride", type=int, default=16, choices=[8, 16])
    a('--method', type=str, default='active')
    a('--loader', type=str, default='region_cityscapes')
    a("--active_method", default='my_random')
    a("--initial_active_method", default='my_random')
    # temperatures / loss weights
    a("--ce_temp", type=float, default=1.0)
    a("--multi_ce_temp", type=float, default=1.0)
    a("--group_ce_temp", type=float, default=1.0)
    a("--coeff", type=float, default=1.0)
    a("--coeff_mc", type=float, default=1.0)
    a("--coeff_gm", type=float, default=1.0)
    a("--loss_type", type=str, default='cross_entropy')
    # active learning
    a("--seed", type=int, default=0)
    a("--start_over", action='store_true', default=False)
    a('--init_checkpoint', type=str, default='checkpoint/resnet50_imagenet_pretrained.tar')
    a('--resume_checkpoint', type=str, default=None)
    a('--datalist_path', type=str, default=None)
    a('--max_iterations', type=int, default=5)
    a('--active_selection_size', type=int, default=100000)
    a('--init_iteration', type=int, default=1)
    a("--cls_weight_coeff", type=float, default=1.0)
    a('--or_labeling', action='store_true', default=False)
    a('--fair_counting', action='store_true', default=False)
    a('--save_scores', action='store_true', default=False)
    a('--num_classes', type=int, default=19)
    a('--nseg', type=int, default=2048)
    a('--cosprop_threshold_method', type=str, default='median')
    # optimisation
    a("--num_workers", type=int, default=4)
    a('--train_batch_size', type=int, default=4)
    a("--weight_decay", type=float, default=1e-5)
    a("--total_itrs", type=int, default=60000)
    a('--finetune_itrs', type=int, default=60000)
    a("--train_lr", type=float, default=0.007)
    a("--cls_lr_scale", type=float, default=10.0)
    a("--optimizer", default='adamw', choices=['adamw', 'sgd'])
    a('--adaptive_train_lr', action='store_true', default=False)
    a("--scheduler", default='poly', choices=['none', 'poly'])
    a("--min_lr", type=float, default=1e-6)
    a("--power", type=float, default=0.9)
    a('--load_optim', action='store_true', default=False)
    a('--ignore_idx', type=int, default=255)
    a('--val_batch_size', type=int, default=4)
    a('--val_num_workers', type=int, default=4)
    a("--set_num_threads", type=int, default=20)
    # bookkeeping
    a('-p', '--model_save_dir', default='./checkpoint/default')
    a('--skip_first_eval', action='store_true', default=False)
    a('--val_start', type=int, default=0)
    a("--val_period", type=int, default=5000)
    a('--log_period', type=int, default=1000)
    a('--dontlog', action='store_true', default=False)
    a('--session_name', default='default')
    a('--val_dataset', default='cityscapes')
    a('--val_data_dir', default='./data/Cityscapes')
    a('--val_datalist', default='dataloader/init_data/cityscapes/val.txt')
    return p
