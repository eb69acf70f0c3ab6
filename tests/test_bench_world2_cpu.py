"""Multi-GPU readiness of bench.py without the hardware: the host logic of the two
legs that shard -- ``timed_scan_round`` (the primary line: ScanRound + barrier / max-over-ranks timing) and ``pool_round_bench`` (the
fixed pool through the selector plugin) -- runs under ``gloo`` with world 2 and the oracle-backed stand-in for the HIP backend, and
must select exactly what the world-1 run selects: same scores, same consumed prefix, same bookkeeping.  So the first real
``--gpus 8`` run cannot fail on host logic (shard offsets, the two exchanges, rank-0 file writes, the tie keys)."""
import importlib.util
import os
import pickle
import socket
import sys
import tempfile
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
B, C, H, W, S, K = 2, 20, 32, 64, 32, 3          # K timed batches per rank at world 2 (6 at world 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helpers import OracleBackend
        bench = _bench()
        dev = torch.device('cpu')
        steps = 2 * K // world
        args = types.SimpleNamespace(batch=B, classes=C, height=H, width=W, nseg=S, steps=steps, warmup=1, nbuf=steps, id_dtype='int64')
        # the same global picture sequence whatever the world size: picture block k of the pool = make_batch(seed_k)
        bufs = [bench.make_batch(17 * (rank * steps + i) + 1, B, C, H, W, S, 'int64', dev) for i in range(steps)]
        timed, n_sel, dt = bench.timed_scan_round(args, dev, rank, world, OracleBackend(), bufs)
        assert dt >= 0 and timed.rnd.plan.n_local == steps * B and timed.n_img == 2 * K * B
        simg, sid, ssc = timed.selected
        res = {'n': int(n_sel), 'scores': timed.scores.numpy().copy(), 'simg': np.asarray(simg)[:int(n_sel)].copy(),
               'sid': np.asarray(sid)[:int(n_sel)].copy(), 'ssc': np.asarray(ssc)[:int(n_sel)].copy(), 'budget': timed.budget}
        # the fixed-pool leg: 11 pictures in batches of 2 (a short last batch; world 2 -> 3 + 3 batches, 6 + 5 pictures)
        save = os.path.join(out_dir, "pool_w%d_r%d" % (world, rank))
        os.makedirs(save)
        pr = bench.pool_round_bench(args, dev, rank, world, False, n_images=11, clicks=60, backend=OracleBackend(), save_dir=save)
        res['pool'] = {k: pr[k] for k in ('regions_selected', 'images_per_rank')}
        fname = os.path.join(save, 'pixbal_selection_01.pkl')
        if rank == 0:
            with open(fname, 'rb') as f:
                res['pool']['consumed'] = pickle.load(f)
            with open(os.path.join(save, 'datalist_01.pkl'), 'rb') as f:
                res['pool']['datalist'] = pickle.load(f)
        else:
            assert not os.path.exists(fname) and not os.path.exists(os.path.join(save, 'datalist_01.pkl'))     # files are rank 0's
        with open(os.path.join(out_dir, "w%d_r%d.pkl" % (world, rank)), "wb") as f:
            pickle.dump(res, f)
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


def test_bench_legs_select_the_same_regions_at_world_two():
    out = tempfile.mkdtemp()
    _run(0, 1, 0, out)
    with open(os.path.join(out, "w1_r0.pkl"), "rb") as f:
        single = pickle.load(f)
    assert single['n'] > 3 and single['pool']['regions_selected'] > 10 and single['pool']['images_per_rank'] == 11
    mp.spawn(_run, args=(2, _free_port(), out), nprocs=2, join=True)
    per_rank = 0
    for r in range(2):
        with open(os.path.join(out, "w2_r%d.pkl" % r), "rb") as f:
            res = pickle.load(f)
        assert res['n'] == single['n'] and res['budget'] == single['budget']
        for k in ('scores', 'simg', 'sid', 'ssc'):
            assert np.array_equal(res[k], single[k]), (k, r)
        assert res['pool']['regions_selected'] == single['pool']['regions_selected']
        per_rank += res['pool']['images_per_rank']
        if r == 0:
            assert res['pool']['consumed'] == single['pool']['consumed']            # the consumed prefix, tuple for tuple
            assert res['pool']['datalist'] == single['pool']['datalist']
    assert per_rank == 11
