#!/usr/bin/env python
"""A `train_AL.py`-shaped active-learning loop on synthetic data (no dataset exists on the GPU box).

The round structure is the reference's (`train_AL.py:37-85`): build the trainer -> load the previous round's best
checkpoint -> select regions (random in round 1, PixBal + ban-ignore afterwards) -> dump the datalist -> train on the
partial labels -> reload the best checkpoint -> evaluate.  Plugins are looked up by name exactly as the reference
does (`importlib.import_module("active_selection." + name)`) after `mulactseg_amd.install_aliases()`.

    python examples/train_AL_synthetic.py --rounds 2 --out /tmp/al_demo
"""
import argparse
import importlib
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mulactseg_amd  # noqa: E402
from mulactseg_amd import synth  # noqa: E402

N_CLS = 19


class SynthScene:
    """Deterministic synthetic 'Cityscapes': image i has a class layout, an image correlated with it, a superpixel map
    and the oracle's multi-hot label per superpixel (classes present in it; the extra column is the undefined bit)."""

    def __init__(self, n_img, H, W, S):
        self.n, self.H, self.W, self.S = n_img, H, W, S
        self.cls = [synth.class_map(1000 + i, H, W, N_CLS, blob=16) for i in range(n_img)]
        self.spx = [synth.superpixel_map(2000 + i, H, W, S) for i in range(n_img)]
        self.multi_hot = np.zeros((n_img, S, N_CLS + 1), dtype=np.uint8)
        for i in range(n_img):
            self.multi_hot[i, self.spx[i].reshape(-1), self.cls[i].reshape(-1)] = 1

    def image(self, i):
        rs = np.random.RandomState(3000 + i)
        palette = np.random.RandomState(7).uniform(-1.5, 1.5, size=(N_CLS, 3)).astype(np.float32)
        return (palette[self.cls[i]].transpose(2, 0, 1) + 0.3 * rs.standard_normal((3, self.H, self.W))).astype(np.float32)


class PoolSet(torch.utils.data.Dataset):
    def __init__(self, scene):
        self.scene = scene
        self.im_idx = [["img/%04d.png" % i, "gt/%04d.png" % i, "spx/%04d.pkl" % i] for i in range(scene.n)]
        self.suppix = {k[2]: sorted(np.unique(scene.spx[i]).tolist()) for i, k in enumerate(self.im_idx)}
        self.isselected = np.zeros((scene.n, scene.S), dtype=np.uint8)

    def __len__(self):
        return len(self.im_idx)

    def __getitem__(self, j):
        i = int(self.im_idx[j][0][4:8])
        return {'images': torch.from_numpy(self.scene.image(i)), 'spx': torch.from_numpy(self.scene.spx[i])}


class LabelSet(torch.utils.data.Dataset):
    def __init__(self, scene):
        self.scene = scene
        self.im_idx, self.suppix = [], {}
        self.multi_hot_cls = scene.multi_hot
        self.id_to_index = {"%04d" % i: i for i in range(scene.n)}

    def __len__(self):
        return len(self.im_idx)

    def __getitem__(self, j):
        i = int(self.im_idx[j][0][4:8])
        spx = self.scene.spx[i]
        return {'images': torch.from_numpy(self.scene.image(i)), 'labels': torch.from_numpy(self.scene.multi_hot[i]),
                'spx': torch.from_numpy(spx), 'spmask': torch.from_numpy(np.isin(spx, self.suppix[self.im_idx[j][2]]))}


class ValSet(torch.utils.data.Dataset):
    def __init__(self, scene):
        self.scene = scene

    def __len__(self):
        return self.scene.n

    def __getitem__(self, i):
        return {'images': torch.from_numpy(self.scene.image(i)), 'labels': torch.from_numpy(self.scene.cls[i].astype(np.int64))}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--images", type=int, default=6)
    ap.add_argument("--size", type=int, nargs=2, default=[64, 96])
    ap.add_argument("--nseg", type=int, default=48)
    ap.add_argument("--budget", type=int, default=40)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--out", default="/tmp/mulactseg_al_demo")
    cli = ap.parse_args(argv)

    mulactseg_amd.install_aliases()
    from utils.common import get_parser, seed_everything          # resolved through the aliases, as in the reference
    from dataloader import RegionActiveDataset, register_dataset_factory
    os.makedirs(cli.out, exist_ok=True)
    args = get_parser().parse_args([
        '-m', 'deeplabv3pluswn_resnet50deepstem', '--separable_conv', '--method', 'active_joint_multi_predignore_lossdecomp',
        '--active_method', 'my_bvsb_predclsbal_pwr_banignore', '--initial_active_method', 'my_random',
        '--ce_temp', '0.1', '--multi_ce_temp', '0.1', '--group_ce_temp', '0.1', '--coeff', '16.0', '--coeff_mc', '8.0', '--coeff_gm', '1.0',
        '--cls_weight_coeff', '6.0', '--or_labeling', '--fair_counting', '--nseg', str(cli.nseg), '--train_batch_size', '2',
        '--val_batch_size', '2', '--num_workers', '0', '--val_num_workers', '0', '--train_lr', '2e-4', '--finetune_itrs', str(cli.iters),
        '--val_period', str(cli.iters), '--log_period', '1', '--active_selection_size', str(cli.budget), '--max_iterations', str(cli.rounds),
        '-p', cli.out])
    args.pretrained_backbone = False
    seed_everything(args.seed)
    logger = logging.getLogger("al")
    scene = SynthScene(cli.images, cli.size[0], cli.size[1], cli.nseg)
    val = ValSet(SynthScene(3, cli.size[0], cli.size[1], cli.nseg))
    register_dataset_factory(lambda a, name, data_root, datalist, imageset: val)
    active_set = RegionActiveDataset(args, PoolSet(scene), LabelSet(scene))
    initial_selector = importlib.import_module("active_selection." + args.initial_active_method).RegionSelector(args)
    active_selector = importlib.import_module("active_selection." + args.active_method).RegionSelector(args)
    Trainer = importlib.import_module("trainer." + args.method.lower())

    history = []
    for selection_iter in range(args.init_iteration, args.max_iterations + 1):
        trainer = Trainer.ActiveTrainer(args, logger, selection_iter)
        active_set.selection_iter = selection_iter
        if selection_iter != 1:
            trainer.load_checkpoint(os.path.join(args.model_save_dir, 'checkpoint%02d.tar' % (selection_iter - 1)))
        selector = initial_selector if selection_iter == 1 else active_selector
        selector.select_next_batch(trainer, active_set, args.active_selection_size)
        active_set.dump_datalist()
        trainer.train(active_set)
        trainer.load_checkpoint(os.path.join(args.model_save_dir, 'checkpoint%02d.tar' % selection_iter))
        table = trainer.eval(selection_iter=selection_iter)
        n_lab = sum(len(v) for v in active_set.trg_label_dataset.suppix.values())
        history.append((selection_iter, n_lab, float(table.split(',')[0])))
        print("[AL %d-round] labelled regions %d, mIoU %.2f" % history[-1], flush=True)
    return history


if __name__ == "__main__":
    main()
