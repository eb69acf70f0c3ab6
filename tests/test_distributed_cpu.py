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
    active.wait_for_writes()
    # every rank holds the same prefix and lists; the FILES are rank 0's business (RegionActiveDataset._writes_files)
    fname = os.path.join(tmp, 'pixbal_selection_01.pkl')
    consumed = None
    if rank == 0:
        with open(fname, 'rb') as f:
            consumed = pickle.load(f)
    else:
        assert not os.path.exists(fname)
    res = dict(scores=scores.numpy(), hist=hist.numpy(), w=sel.cls_weight.numpy(), cum=sel.cumulated_pred_prob,
               consumed=consumed, isselected=pool.isselected, n_local=sel._round.plan.n_local,
               labelled=dict(label.suppix), label_idx=list(label.im_idx), pooled=dict(pool.suppix))
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
        for k in ('labelled', 'label_idx', 'pooled'):
            assert res[k] == single[k], (k, r)
        assert res['consumed'] == (single['consumed'] if r == 0 else None)
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


# ---------------------------------------------------------------------------------------------------------------------
# data-parallel training: global loss normalisers and decorrelated data order
# ---------------------------------------------------------------------------------------------------------------------
def _loss_inputs():
    sys.path.insert(0, ROOT)
    from mulactseg_amd import synth
    N, C, H, W, S = 4, 20, 24, 32, 24
    z = synth.logits(31, N, C, H, W)
    sm = [synth.train_crop(40 + i, H, W, S, frac_selected=0.15 + 0.15 * i) for i in range(N)]     # very different counts per picture
    spx, msk = np.stack([a for a, _ in sm]), np.stack([b for _, b in sm])
    tgt = np.stack([synth.multi_hot_targets(60 + i, S, C) for i in range(N)])
    return z, tgt, spx, msk, S


def _run_loss(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helpers import OracleLossOps
        from mulactseg_amd.utils import loss as loss_mod
        loss_mod.ops = OracleLossOps                      # the scans come from oracle/exact.c; everything else is the product code
        z, tgt, spx, msk, S = _loss_inputs()
        per = z.shape[0] // world
        sl = slice(rank * per, (rank + 1) * per)
        zt = torch.from_numpy(z[sl].copy()).requires_grad_(True)
        crit = loss_mod.FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=True)
        group, ce, mc = crit(zt, torch.from_numpy(tgt[sl]), torch.from_numpy(spx[sl]), torch.from_numpy(msk[sl]))
        loss = (16.0 * ce + 8.0 * mc + 1.0 * group) * world          # what the trainers do before DDP averages the gradients
        loss.backward()
        g = zt.grad.clone()
        if world > 1:
            # DistributedDataParallel averages parameter gradients over ranks; for the logits themselves that is dz / world
            g = g / world
        torch.save(dict(losses=torch.stack([group, ce, mc]).detach(), grad=g, acc=crit.last_acc.clone()),
                   os.path.join(out_dir, "loss_w%d_r%d.pt" % (world, rank)))
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


def test_two_ranks_with_batch_two_optimise_the_single_rank_objective_of_batch_four():
    """FusedPartialLabelLoss(sync_normalisers=True): the integer sums / counts are all-reduced before the division, so both ranks
    hold the loss of the GLOBAL batch (bit-identical to one rank with all four pictures), and loss * world followed by DDP's
    gradient averaging gives every picture exactly the single-rank gradient."""
    out = tempfile.mkdtemp()
    _run_loss(0, 1, 0, out)
    single = torch.load(os.path.join(out, "loss_w1_r0.pt"))
    mp.spawn(_run_loss, args=(2, _free_port(), out), nprocs=2, join=True)
    parts = [torch.load(os.path.join(out, "loss_w2_r%d.pt" % r)) for r in range(2)]
    for p in parts:
        assert torch.equal(p['losses'].view(torch.int32), single['losses'].view(torch.int32))      # same bits on every rank
        assert torch.equal(p['acc'], single['acc'])
    grad = torch.cat([p['grad'] for p in parts])
    assert torch.equal(grad.view(torch.int32), single['grad'].view(torch.int32))
    assert float(single['losses'].abs().sum()) > 0 and float(single['grad'].abs().sum()) > 0


def _run_seeds(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import random
        from mulactseg_amd.trainer.base import BaseTrainer
        t = types.SimpleNamespace(args=types.SimpleNamespace(seed=1), selection_iter=2)
        seed = BaseTrainer.loader_seed(t)
        random.seed(1)                                   # the process-global stream the selectors use stays identical
        with open(os.path.join(out_dir, "seed%d.pkl" % rank), "wb") as f:
            pickle.dump((seed, random.random()), f)
    finally:
        dist.destroy_process_group()


def test_data_order_seeds_differ_per_rank_while_the_global_stream_agrees():
    out = tempfile.mkdtemp()
    mp.spawn(_run_seeds, args=(2, _free_port(), out), nprocs=2, join=True)
    (s0, g0), (s1, g1) = [pickle.load(open(os.path.join(out, "seed%d.pkl" % r), "rb")) for r in range(2)]
    assert s0 != s1 and g0 == g1


def _split_default_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from mulactseg_amd import ops
    before = ops.sk_split_default()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inside = ops.sk_split_default()
        os.environ["MAS_SK_SPLIT"] = "on"
        forced = ops.sk_split_default()
        del os.environ["MAS_SK_SPLIT"]
    finally:
        dist.destroy_process_group()
    q.put((rank, before, inside, forced))


def test_stream_k_hand_off_is_off_by_default_under_more_than_one_rank():
    """ops.sk_split_default: tiles are split over workgroups in a single-GPU process only; with world_size > 1 the whole-tile plan
    (no hand-off, nothing can give up and leave the other ranks in the gradient all-reduce) unless MAS_SK_SPLIT=on."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_default_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert got == [(0, True, False, True), (1, True, False, True)]
