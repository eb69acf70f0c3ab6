"""ResidentRegionDataset + RegionActiveDataset bookkeeping on the CPU (no kernels are launched: the datasets only
hold references to the pictures until a sample is drawn)."""
import types

import numpy as np
import torch

from mulactseg_amd.dataloader import RegionActiveDataset
from mulactseg_amd.dataloader.resident import ResidentRegionDataset


def _sets(tmp_path, n=3, nseg=16, C=20):
    args = types.SimpleNamespace(nseg=nseg, ignore_idx=255, fair_counting=True, or_labeling=True, model_save_dir=str(tmp_path),
                                 finetune_itrs=1)
    pics = [torch.zeros((8, 8, 3), dtype=torch.uint8) for _ in range(n)]
    spxs = [torch.zeros((8, 8), dtype=torch.int64) for _ in range(n)]
    rs = np.random.RandomState(0)
    mh = (rs.rand(n, nseg, C) < 0.1).astype(np.uint8)
    mh[..., 0] = 1
    names = [("im/a_%d.png" % k, "lb/a_%d.png" % k, "sp/a_%d.pkl" % k) for k in range(n)]
    start = {names[0][2]: [0, 1]}                                   # image 0 starts with two labelled regions
    pool_dict = {nm[2]: [i for i in range(nseg) if not (k == 0 and i in (0, 1))] for k, nm in enumerate(names)}
    label = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-label', region_dict=start)
    pool = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-ulabel', region_dict=pool_dict)
    return args, names, mh, RegionActiveDataset(args, pool, label)


def test_expand_training_set_accumulates_on_an_already_labelled_image(tmp_path):
    """entries of im_idx are lists, so a second selection from a labelled image appends to its id list
    instead of overwriting it, no duplicate image entry appears, and an emptied pool image is removed."""
    args, names, mh, aset = _sets(tmp_path)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    assert label.im_idx == [list(names[0])] and label.suppix[names[0][2]] == [0, 1]
    order = [(0.9, ','.join(names[0]), 5), (0.8, ','.join(names[1]), 3), (0.7, ','.join(names[0]), 7), (0.6, ','.join(names[1]), 4)]
    n = aset.expand_training_set(order, 10 ** 6, 'x')
    assert n == 4
    assert label.suppix[names[0][2]] == [0, 1, 5, 7]
    assert label.suppix[names[1][2]] == [3, 4]
    assert label.im_idx == [list(names[0]), list(names[1])]          # no duplicates, lists as in the reference
    assert 5 not in pool.suppix[names[0][2]] and 7 not in pool.suppix[names[0][2]]
    assert pool.isselected[0, 5] == 1 and pool.isselected[1, 4] == 1 and pool.isselected.sum() == 4
    # empty one pool image completely: its key and its id list disappear
    rest = [(0.5, ','.join(names[2]), i) for i in range(args.nseg)]
    aset.expand_training_set(rest, 10 ** 6, 'x')
    assert names[2][2] not in pool.suppix and list(names[2]) not in pool.im_idx
    assert len(label) == 3 and len(pool) == 2


def test_budget_is_the_click_cost_under_fair_counting(tmp_path):
    args, names, mh, aset = _sets(tmp_path)
    order = [(1.0 - 0.01 * i, ','.join(names[1]), i) for i in range(args.nseg)]
    costs = np.cumsum([int(mh[1, i].sum()) for i in range(args.nseg)])
    budget = int(costs[4])                                            # reached exactly by 5 regions -> one more is taken
    n = aset.expand_training_set(order, budget, 'x')
    assert n == int(np.searchsorted(costs, budget, side='right')) + 1
