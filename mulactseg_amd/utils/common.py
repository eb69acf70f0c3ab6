"""Flags, seeding, logging set-up and running averages the drivers and hot-path plugins read -- the in-scope subset of the
reference's ``utils/common.py`` (``get_parser`` :208-370 carries ~110 flags, most of them for ablation
trainers that are out of scope; every flag below keeps the reference's name and default), plus the three functions
its drivers call before the round loop (``initialization`` :133-142, ``preprocess`` :167-203, ``arg_assert`` :205-231)."""
import argparse
import logging
import os
import random
import sys

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
    # data layer (dataloader.get_active_dataset and the file-backed loaders)
    a("--active_mode", default='region', choices=['scan', 'region'])
    a('--src_dataset', default='cityscapes', choices=['cityscapes', 'voc', 'GTA5', 'SYNTHIA'])
    a('--src_data_dir', default='./data/Cityscapes')
    a('--trg_dataset', default='cityscapes')
    a('--trg_data_dir', default='./data/Cityscapes')
    a('--trg_datalist', default='dataloader/init_data/cityscapes/train_seed2048.txt')
    a('--region_dict', default='dataloader/init_data/cityscapes/train_seed2048.dict')
    a('--train_transform', default=None)
    a('--prob_dominant', action='store_true', default=False)
    a("--known_ignore", action='store_true', default=False)
    a('--dominant_labeling', action='store_true', default=False)
    a('--spx_method', type=str, default="seeds", choices=["seeds", "slic"])
    a('--nseg_list', nargs='+', default=None, type=int)
    a('--plbl_type', type=str, default=None)
    a('--loading', default='binary', choices=['binary', 'naive', 'tensor'])
    a('--ignore_size', type=int, default=0)
    a('--mark_topk', type=int, default=-1)
    a('--stage2', action='store_true', default=False)
    a('--load_smaller_spx', action='store_true', default=False)
    a('--small_nseg', type=int, default=2048)
    a('--trim_kernel_size', type=int, default=3)
    a('--trim_multihot_boundary', action='store_true', default=False)
    a('--wandb_tags', nargs='+', default=None)
    a('--wandb_group', default=None)
    return p


def initialize_logging(model_save_dir):
    """The run directory (+ ``AL_record/``) and the file logger ``log_train.txt`` -- ``utils/common.py:69-82``."""
    os.makedirs(os.path.join(model_save_dir, "AL_record"), exist_ok=True)
    fname = os.path.join(model_save_dir, 'log_train.txt')
    fmt, datefmt = '%(asctime)s %(levelname)s: %(message)s', '%Y%m%d %H:%M:%S'
    root = logging.getLogger()
    if root.handlers:       # basicConfig is a no-op once the root logger has handlers (a host application, a test runner): add the file
        if not any(isinstance(h, logging.FileHandler) and getattr(h, 'baseFilename', None) == os.path.abspath(fname) for h in root.handlers):
            h = logging.FileHandler(fname)
            h.setFormatter(logging.Formatter(fmt, datefmt))
            root.addHandler(h)
        root.setLevel(logging.DEBUG)
    else:
        logging.basicConfig(level=logging.DEBUG, format=fmt, datefmt=datefmt, filename=fname)
    logger = logging.getLogger("Trainer")
    logger.info("%s New Experiment %s" % ('-' * 20, '-' * 20))
    logging.getLogger('PIL').setLevel(logging.INFO)
    return logger


def initialization(args):
    """Seed every generator, open the log, record the command line and the flags -- ``utils/common.py:133-142``."""
    seed_everything(args.seed)
    logger = initialize_logging(args.model_save_dir)
    logger.info(' '.join(sys.argv))
    logger.info(args)
    return logger


def gen_save_name(args):
    """The stage-1 run directory carries the experiment's key flags in its name -- ``utils/common.py:144-155``."""
    args.model_save_dir = '{}_{}_sp{}_nlbl{}k_iter{}k_method-{}-_coeff{}_ign{}_lr{}_'.format(
        args.model_save_dir, args.active_method, args.nseg, float(args.active_selection_size) / 1000, float(args.finetune_itrs) / 1000,
        args.method, args.coeff, args.known_ignore, args.train_lr)


def avoid_duplication(args):
    """An existing run directory is not reused: a trailing digit counts up, otherwise ``_1`` is appended (:157-165)."""
    while os.path.exists(args.model_save_dir) and 'naive' not in args.model_save_dir:
        d = str(args.model_save_dir)
        args.model_save_dir = '%s%d' % (d[:-1], int(d[-1]) + 1) if d[-1].isnumeric() else d + "_1"


def preprocess(args):
    """Derived flags -- ``utils/common.py:167-203``: the largest of ``--nseg_list`` is THE nseg (the crop pads id maps with it), session
    names, the stage-1 run directory, and the datalist / region dictionary that belong to ``--nseg`` and the labelling mode."""
    if args.nseg_list is not None:
        args.nseg = args.nseg_list[-1]
    tail = args.model_save_dir.split('/')[-1]
    args.session_id = tail
    args.session_name = '{}_{}'.format(args.method, tail)
    if not args.stage2:
        gen_save_name(args)
        avoid_duplication(args)
    if str(args.nseg) not in args.trg_datalist:
        args.trg_datalist = "dataloader/init_data/cityscapes/train_seed{}.txt".format(args.nseg)
    if str(args.nseg) not in args.region_dict:
        args.region_dict = 'dataloader/init_data/cityscapes/train_seed{}.dict'.format(args.nseg)
    if args.dominant_labeling and 'dominant' not in args.trg_datalist:
        args.trg_datalist = '{}_dominant.txt'.format(args.trg_datalist.split('.')[0])
    if args.or_labeling and 'or' not in args.trg_datalist:
        args.trg_datalist = '{}_or.txt'.format(args.trg_datalist.split('.')[0])
    if args.known_ignore:
        assert 'ignore' in args.loader


def arg_assert(args):
    """Consistency of the flags -- ``utils/common.py:205-231``."""
    assert args.init_checkpoint is not None
    assert str(args.nseg) in args.trg_datalist
    assert str(args.nseg) in args.region_dict
    if args.dominant_labeling:
        assert 'dominant' in args.trg_datalist
        assert "_or_" not in args.loader.lower()
    if args.or_labeling:
        assert 'or' in args.trg_datalist
    if (args.datalist_path is not None or args.resume_checkpoint is not None) and not args.stage2:
        assert args.datalist_path.split('/')[-2] == args.resume_checkpoint.split('/')[-2]       # both from the same run directory
    if 'deeplabv3pluswn_resnet50' in args.model and args.ce_temp == 1:
        print("Check CE temp: {}".format(args.ce_temp))
    assert args.ignore_size == 0 and args.mark_topk == -1            # deprecated options


def worker_init_fn(worker_id):
    np.random.seed(worker_id)
    random.seed(worker_id)
    torch.manual_seed(worker_id)
