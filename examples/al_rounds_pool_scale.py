#!/usr/bin/env python
"""BASELINE.json config 5 as a timed loop on ONE MI355X: the reference's round structure (`train_AL.py:37-85`) over a resident,
Cityscapes-sized synthetic pool -- 2 975 pictures x 2 048 superpixels, 100 000 clicks per round, 5 rounds -- with a per-round
wall-clock breakdown written as JSON.  (mIoU against the reference needs the dataset, which the GPU box does not have; what
this run pins is that the whole loop works at pool scale and what each phase costs.)

Per round: trainer construction + previous checkpoint -> selection (round 1 random, then PixBal + ban-ignore: model forward
over the whole pool + single-pass scan + K4 on 6.09 M keys) -> datalist dump -> stage-1 training on the partial labels
(resident pictures, device augmentation, fused losses from the quarter-resolution logits) -> checkpoint reload -> validation.
After the last round the stage-2 generator (K9) runs over `--stage2-images` labelled pictures.

Checked while it runs: the labelled set grows by exactly the consumed prefix of every round (the selection pickle), no region is
labelled twice, the pool shrinks by the same regions, and `datalist_RR.pkl` reloads to the same lists (`train_AL.py:43-57`).

    python examples/al_rounds_pool_scale.py --json gpurun_out/al_rounds.json
    python examples/al_rounds_pool_scale.py --images 48 --height 256 --width 512 --nseg 256 --budget 600 --iters 6 --val-images 4   # small
"""
import argparse
import importlib
import json
import logging
import os
import pickle
import random
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mulactseg_amd  # noqa: E402

N_CLS = 19


class ResidentVal(torch.utils.data.Dataset):
    """Validation pictures and label maps resident on the device (synthetic)."""
    device_resident = True

    def __init__(self, n, H, W, dev, seed=99):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        self.x = torch.randn((n, 3, H, W), generator=g, device=dev)
        self.y = torch.randint(0, N_CLS, (n, H // 32 + 1, W // 32 + 1), generator=g, device=dev).repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :H, :W].contiguous()

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {'images': self.x[i], 'labels': self.y[i]}


def build_pool(args, n_img, H, W, S, dev, chunk=64):
    from mulactseg_amd.dataloader.resident import ResidentRegionDataset
    from mulactseg_amd.synth_pool import device_superpixel_maps
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pics = torch.empty((n_img, H, W, 3), dtype=torch.uint8, device=dev)
    maps = torch.empty((n_img, H, W), dtype=torch.int16, device=dev)
    for lo in range(0, n_img, chunk):
        hi = min(lo + chunk, n_img)
        pics[lo:hi] = torch.randint(0, 256, (hi - lo, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
        maps[lo:hi] = device_superpixel_maps([7 * 100003 + i for i in range(lo, hi)], H, W, S, dev, torch.int16)
    rs = np.random.RandomState(11)
    k = rs.choice(4, size=(n_img, S), p=[0.70, 0.22, 0.06, 0.02]) + 1           # classes under a region (click cost)
    first = rs.randint(0, N_CLS + 1, size=(n_img, S))
    mh = np.zeros((n_img, S, N_CLS + 1), dtype=np.uint8)
    for j in range(4):
        np.put_along_axis(mh, ((first + j) % (N_CLS + 1))[..., None], (k > j)[..., None].astype(np.uint8), axis=2)
    np.put_along_axis(mh, first[..., None], 1, axis=2)
    mh_dev = torch.from_numpy(mh).to(dev)
    names = [("leftImg8bit/city_%05d.png" % i, "gtFine/city_%05d.png" % i, "superpixel/city_%05d.pkl" % i) for i in range(n_img)]
    pool = ResidentRegionDataset(args, list(pics), list(maps), mh_dev, names, split='active-ulabel')
    label = ResidentRegionDataset(args, list(pics), list(maps), mh_dev, names, split='active-label', region_dict={}, rng=random.Random(5))
    label.transform.size = (args.crop, args.crop)
    pool.initial_valid_table = lambda: np.ones((n_img, S), dtype=np.uint8)        # every id listed at the start
    return pool, label, mh


def numpy_prefix(scores, valid, rank, cost, budget):
    """The reference's ``sorted(tuples, reverse=True)`` + budget walk on arrays (active_selection/base.py:37,
    dataloader/region_active_dataset.py:31-73): descending (score, path rank, id), stop after the region that makes the click cost
    exceed the budget.  Returns (picture row, region id) of the consumed prefix."""
    n, s = scores.shape
    img = np.repeat(np.arange(n), s)
    rid = np.tile(np.arange(s), n)
    keep = valid.reshape(-1) != 0
    img, rid, sc = img[keep], rid[keep], scores.reshape(-1)[keep]
    order = np.lexsort((-rid, -rank[img], -sc.astype(np.float64)))
    img, rid = img[order], rid[order]
    cum = np.cumsum(cost[img, rid].astype(np.int64))
    over = np.nonzero(cum > budget)[0]
    m = len(img) if len(over) == 0 else int(over[0]) + 1
    return img[:m], rid[:m]


def region_set(suppix):
    return {(k, i) for k, ids in suppix.items() for i in ids}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--images", type=int, default=2975)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--nseg", type=int, default=2048)
    ap.add_argument("--budget", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--crop", type=int, default=768)
    ap.add_argument("--val-images", type=int, default=32)
    ap.add_argument("--stage2-images", type=int, default=32)
    ap.add_argument("--out", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--check-order", action="store_true",
                    help="every scored round: the consumed prefix must equal a numpy lexsort restatement of the reference's tuple sort + walk")
    cli = ap.parse_args(argv)
    out_dir = cli.out or tempfile.mkdtemp(prefix="mas_al_pool_")
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device('cuda:0')

    mulactseg_amd.install_aliases()
    from utils.common import get_parser, seed_everything          # resolved through the aliases, as in the reference
    from dataloader import RegionActiveDataset, register_dataset_factory
    args = get_parser().parse_args([
        '-m', 'deeplabv3pluswn_resnet50deepstem', '--separable_conv', '--method', 'active_joint_multi_predignore_lossdecomp',
        '--active_method', 'my_bvsb_predclsbal_pwr_banignore', '--initial_active_method', 'my_random',
        '--ce_temp', '0.1', '--multi_ce_temp', '0.1', '--group_ce_temp', '0.1', '--coeff', '16.0', '--coeff_mc', '8.0', '--coeff_gm', '1.0',
        '--cls_weight_coeff', '6.0', '--or_labeling', '--fair_counting', '--nseg', str(cli.nseg), '--train_batch_size', '4',
        '--val_batch_size', '4', '--num_workers', '0', '--val_num_workers', '0', '--train_lr', '2e-5', '--finetune_itrs', str(cli.iters),
        '--val_period', str(cli.iters), '--log_period', str(max(1, cli.iters // 4)), '--active_selection_size', str(cli.budget),
        '--max_iterations', str(cli.rounds), '-p', out_dir])
    args.pretrained_backbone = False
    args.crop = cli.crop
    seed_everything(args.seed)
    logger = logging.getLogger("al")

    t0 = time.perf_counter()
    pool, label, mh = build_pool(args, cli.images, cli.height, cli.width, cli.nseg, dev)
    val = ResidentVal(cli.val_images, cli.height, cli.width, dev)
    register_dataset_factory(lambda a, name, data_root, datalist, imageset: val)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    active_set = RegionActiveDataset(args, pool, label)
    initial_selector = importlib.import_module("active_selection." + args.initial_active_method).RegionSelector(args)
    active_selector = importlib.import_module("active_selection." + args.active_method).RegionSelector(args)
    kept = {}
    if cli.check_order:                      # keep the score tensor and the pool's valid table of every scored round
        inner = active_selector.calculate_scores_tensor

        def keep_scores(trainer, pool_set, want_hist=False):
            kept['valid'] = np.array(active_selector._pool_valid(active_set, pool_set), copy=True)
            kept['rows'] = [row_of_all[k[2]] for k in pool_set.im_idx]
            kept['scores'] = inner(trainer, pool_set, want_hist)
            return kept['scores']
        active_selector.calculate_scores_tensor = keep_scores
    row_of_all = {n[2]: i for i, n in enumerate(pool.im_idx)}
    Trainer = importlib.import_module("trainer." + args.method.lower())
    cost = mh.sum(axis=2)
    pool_names = [list(n) for n in pool.im_idx]
    row_of = {n[2]: i for i, n in enumerate(pool.im_idx)}

    def clock():
        torch.cuda.synchronize()
        return time.perf_counter()

    rounds = []
    labelled_before = set()
    n_pool0 = cli.images * cli.nseg
    for selection_iter in range(args.init_iteration, args.max_iterations + 1):
        r = {"round": selection_iter}
        t = clock()
        trainer = Trainer.ActiveTrainer(args, logger, selection_iter)
        active_set.selection_iter = selection_iter
        if selection_iter != 1:
            trainer.load_checkpoint(os.path.join(args.model_save_dir, 'checkpoint%02d.tar' % (selection_iter - 1)))
        r["build_trainer_and_load_checkpoint_s"] = clock() - t
        t = clock()
        selector = initial_selector if selection_iter == 1 else active_selector
        selector.select_next_batch(trainer, active_set, args.active_selection_size)
        r["selection_s"] = clock() - t
        t = clock()
        active_set.dump_datalist()
        r["dump_datalist_s"] = clock() - t

        # -- bookkeeping invariants of the round (train_AL.py:43-57, region_active_dataset.py:31-73) ---------------------------
        with open(os.path.join(args.model_save_dir, '%s_selection_%02d.pkl' % (selector.active_method, selection_iter)), 'rb') as f:
            consumed = pickle.load(f)
        new = {(p.split(',')[2], i) for _, p, i in consumed}
        labelled = region_set(label.suppix)
        assert len(new) == len(consumed), "a region was selected twice in one round"
        assert labelled == labelled_before | new and not (labelled_before & new), "labelled set != previous + consumed prefix"
        assert sum(len(v) for v in pool.suppix.values()) == n_pool0 - len(labelled), "pool did not shrink by the labelled regions"
        clicks = int(sum(cost[row_of[k], i] for k, i in new))
        last = consumed[-1]
        assert clicks > args.active_selection_size >= clicks - int(cost[row_of[last[1].split(',')[2]], last[2]]), "budget walk stopped at the wrong region"
        with open(os.path.join(args.model_save_dir, 'datalist_%02d.pkl' % selection_iter), 'rb') as f:
            dl = pickle.load(f)
        assert dl['trg_label_suppix'] == label.suppix and dl['trg_pool_im_idx'] == pool.im_idx
        if cli.check_order and selection_iter != 1:
            sc = kept['scores'].cpu().numpy()
            rows = np.asarray(kept['rows'])
            paths = [','.join(pool_names[i]) for i in rows]
            rank = np.empty(len(paths), dtype=np.int64)
            rank[np.argsort(np.array(paths))] = np.arange(len(paths))
            ni, nr = numpy_prefix(sc, kept['valid'], rank, cost[rows], args.active_selection_size)
            want = [(pool_names[rows[i]][2], int(j)) for i, j in zip(ni, nr)]
            got = [(p.split(',')[2], int(i)) for _, p, i in consumed]
            assert got == want, "consumed prefix differs from the lexsort restatement (first mismatch at %d)" % next(
                (k for k, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
            r["order_checked_regions"] = len(want)
        labelled_before = labelled
        r.update(regions_selected=len(new), clicks=clicks, labelled_regions_total=len(labelled), labelled_pictures=len(label.im_idx))

        t = clock()
        trainer.train(active_set)
        r["training_s"] = clock() - t
        r["train_iterations"] = int(args.finetune_itrs)
        # (training_s includes the validation pass and the checkpoint write that trainer.train() does at its last iteration)
        r["train_images_per_s"] = 4 * int(args.finetune_itrs) / r["training_s"]
        t = clock()
        trainer.load_checkpoint(os.path.join(args.model_save_dir, 'checkpoint%02d.tar' % selection_iter))
        table = trainer.eval(selection_iter=selection_iter)
        r["reload_best_and_validation_s"] = clock() - t
        r["val_miou_synthetic"] = float(table.split(',')[0])
        r["round_total_s"] = sum(v for k, v in r.items() if k.endswith('_s') and k != 'train_images_per_s')
        rounds.append(r)
        print("[AL %d-round] %s" % (selection_iter, json.dumps(r)), flush=True)

    # -- stage 2: pseudo labels for the first K labelled pictures with the final model (K9) ----------------------------------
    from mulactseg_amd import ops
    from mulactseg_amd.dataloader.formats import selection_lut
    stage2 = None
    if cli.stage2_images > 0:
        net = trainer.net.eval()
        keys = label.im_idx[:cli.stage2_images]
        t = clock()
        frac = []
        with torch.no_grad():
            for key in keys:
                k = row_of[key[2]]
                item = pool.__getpoolitem__(k)
                spx = item['spx'].to(torch.int64)[None]
                lut = selection_lut(label.suppix[key[2]], cli.nseg, dev)
                msk = lut[spx.clamp(min=0, max=cli.nseg)]
                feats, logits = net.feat_forward_lowres(item['images'][None])
                plbl = ops.stage2_pseudo_labels(feats.contiguous(), logits.contiguous(), item['labels'][None].contiguous(), msk.contiguous(),
                                                spx.contiguous(), True)
                frac.append(float((plbl != 255).float().mean()))
        dt = clock() - t
        stage2 = {"pictures": len(keys), "seconds": dt, "ms_per_picture": dt / max(1, len(keys)) * 1e3, "mean_labelled_fraction": float(np.mean(frac))}

    report = {"workload": "5-round active-learning loop (train_AL.py:37-85) on one MI355X, synthetic resident pool",
              "pool": {"pictures": cli.images, "superpixels_per_picture": cli.nseg, "size": [cli.height, cli.width], "regions": n_pool0},
              "budget_clicks_per_round": cli.budget, "train_iterations_per_round": cli.iters, "train_batch": [4, 3, cli.crop, cli.crop],
              "validation_pictures": cli.val_images, "setup_s (synthetic pool generation, not part of a round)": setup_s,
              "rounds": rounds, "stage2_generation": stage2, "total_rounds_s": sum(r["round_total_s"] for r in rounds),
              "invariants_checked": "labelled set == previous + consumed prefix, no duplicates, pool shrinks by the same regions, budget walk stops "
                                    "after the first region exceeding the budget, datalist_RR.pkl reloads to the same lists"}
    if cli.json:
        with open(cli.json, "w") as f:
            json.dump(report, f, indent=1)
    print(json.dumps(report))
    return report


if __name__ == "__main__":
    main()
