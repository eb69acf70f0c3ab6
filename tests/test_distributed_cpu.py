"""N > 1 path on CPU: world_size-2 (and 3, ragged) ``gloo`` runs of the sharded acquisition round must
reproduce the single-process result bit for bit (scores, histograms, class weights, selection).
The GPU backend is replaced by the oracle-backed stand-in of tests/helpers.py; what is under test is
the sharding in whole reference batches, the two exchanges and the replicated merge."""
import os
import pickle
import socket
import sys
import tempfile
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    from mulactseg_amd import synth
    n_img, C, H, W, S = 7, 20, 24, 40, 32                    # 7 images / batch 2 -> 4 batches, short last one
    z = synth.logits(77, n_img, C, H, W)
    z[1, C - 1, :12, :20] += 1.5
    spx = np.stack([synth.superpixel_map(500 + i, H, W, S) for i in range(n_img)])
    im_idx = [["i/%03d.png" % ((i * 5) % n_img), "l/%03d.png" % i, "s/spx_%04d.pkl" % i] for i in range(n_img)]
    suppix = {k[2]: sorted(set(np.unique(spx[i]).tolist()) - {3, 7}) for i, k in enumerate(im_idx)}
    mh = np.stack([synth.multi_hot_targets(900 + i, S, C) for i in range(n_img)])
    return z, spx, im_idx, suppix, mh, S


def _run(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from helpers import FakePool, OracleBackend, fake_trainer, selector_args
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd.dataloader import RegionActiveDataset
    z, spx, im_idx, suppix, mh, S = _inputs()
    tmp = tempfile.mkdtemp()
    args = selector_args(val_batch_size=2, nseg=S, model_save_dir=tmp, active_method='pixbal')
    pool = FakePool(z, spx, im_idx, suppix)
    pool.isselected = np.zeros((len(im_idx), S), dtype=np.uint8)
    label = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=mh,
                                  id_to_index={"spx_%04d" % i: i for i in range(len(im_idx))})
    active = RegionActiveDataset(args, pool, label)
    active.selection_iter = 1
    sel = banignore.RegionSelector(args)
    sel.backend = OracleBackend()
    scores, hist = sel.calculate_scores_tensor(fake_trainer(), pool, want_hist=True)
    sel.select_next_batch(fake_trainer(save_dir=tmp), active, 30)
    with open(os.path.join(tmp, 'pixbal_selection_01.pkl'), 'rb') as f:
        consumed = pickle.load(f)
    res = dict(scores=scores.numpy(), hist=hist.numpy(), w=sel.cls_weight.numpy(), cum=sel.cumulated_pred_prob,
               consumed=consumed, isselected=pool.isselected, n_local=sel._round.plan.n_local)
    with open(os.path.join(out_dir, "w%d_r%d.pkl" % (world, rank)), "wb") as f:
        pickle.dump(res, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_round_equals_single_process(world):
    out = tempfile.mkdtemp()
    _run(0, 1, 0, out)
    with open(os.path.join(out, "w1_r0.pkl"), "rb") as f:
        single = pickle.load(f)
    port = _free_port()
    mp.spawn(_run, args=(world, port, out), nprocs=world, join=True)
    n_local = 0
    for r in range(world):
        with open(os.path.join(out, "w%d_r%d.pkl" % (world, r)), "rb") as f:
            res = pickle.load(f)
        n_local += res['n_local']
        for k in ('scores', 'hist', 'w', 'cum', 'isselected'):
            assert np.array_equal(res[k], single[k]), (k, r)
        assert res['consumed'] == single['consumed']
    assert n_local == 7                                     # every image scored exactly once


def test_shard_plan_covers_pool_in_whole_batches():
    from mulactseg_amd.active_selection.engine import ShardPlan
    for n_img, bs, world in [(2975, 4, 8), (7, 2, 3), (5, 4, 8), (1464, 12, 4), (1, 4, 2)]:
        seen = []
        for r in range(world):
            p = ShardPlan(n_img, bs, r, world)
            assert p.img_lo % bs == 0 or p.img_lo == n_img
            seen += p.local_indices
            assert p.n_local <= p.per_rank_imgs
        assert seen == list(range(n_img))


def _run_meter(rank, world, port, out_dir):
    import torch.distributed as dist
    from mulactseg_amd.utils.miou import MeanIoU
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        m = MeanIoU(3, 255)
        m._before_epoch()
        if rank == 0:                                   # rank 1 saw no batch at all
            m._ensure(torch.device('cpu'))[:] = torch.arange(12)
        m.all_reduce(torch.device('cpu'))
        torch.save(m._counts.clone(), os.path.join(out_dir, "meter%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_iou_counters_sum_over_ranks_even_when_a_rank_saw_nothing():
    out = tempfile.mkdtemp()
    mp.spawn(_run_meter, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(os.path.join(out, "meter0.pt")), torch.load(os.path.join(out, "meter1.pt"))
    assert torch.equal(a, torch.arange(12)) and torch.equal(a, b)
