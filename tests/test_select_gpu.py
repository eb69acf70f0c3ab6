"""GPU parity of K4 (device ordering + budget walk) against the reference's semantics: Python
``sorted(tuples, reverse=True)`` over (score, "img,lbl,spx", id) and the walk of
``RegionActiveDataset.expand_training_set`` -- restated in oracle/port.py and pinned to the executed
reference by tests/golden/g1_pixbal_city.npz.  Index outputs are bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    return ops


def _device_select(ops, scores, valid, paths, cost_bits, budget, max_out=None):
    img_rank, img_of_rank = ops.path_ranks(paths)
    S = scores.shape[1]
    keys = ops.region_keys(torch.from_numpy(scores).cuda(), None if valid is None else torch.from_numpy(valid).cuda(),
                           torch.from_numpy(img_rank).cuda())
    skeys = ops.sort_keys_desc(keys)
    cb = None if cost_bits is None else torch.from_numpy(cost_bits.astype(np.uint8)).cuda()
    nsel, simg, sid, ssc = ops.budget_walk(skeys, cb, torch.from_numpy(img_of_rank).cuda(), S, budget, max_out)
    n = int(nsel.item())
    return n, simg.cpu().numpy(), sid.cpu().numpy(), ssc.cpu().numpy()


def _python_select(scores, valid, paths, cost, budget):
    from oracle import port
    tuples = []
    for i, p in enumerate(paths):
        for s in range(scores.shape[1]):
            if valid is None or valid[i, s]:
                tuples.append((float(scores[i, s]), p, s))
    idx = {p: i for i, p in enumerate(paths)}
    cost_fn = None if cost is None else (lambda path, rid: int(cost[idx[path], rid]))
    return [(idx[p], s, sc) for sc, p, s in port.select_regions(tuples, budget, cost_fn)]


def test_g1_selection_matches_executed_reference():
    ops = _gpu()
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    n_img, S = int(g['n_img']), int(g['S'])
    scores = g['scores_tensor'].astype(np.float32)
    valid = np.zeros((n_img, S), dtype=np.uint8)
    valid[g['list_img'], g['list_id']] = 1
    paths = ["leftImg8bit/train/c/img_%04d.png,gtFine/train/c/lbl_%04d.png,superpixel/train/c/spx_%04d.pkl" % (i, i, i)
             for i in range(n_img)]
    mh = g['multi_hot']
    n, simg, sid, ssc = _device_select(ops, scores, valid, paths, mh.sum(axis=2), int(g['budget']))
    assert n == len(g['consumed_img'])
    assert np.array_equal(simg[:n], g['consumed_img']) and np.array_equal(sid[:n], g['consumed_id'])
    assert np.array_equal(ssc[:n].astype(np.float64), g['consumed_score'])
    assert np.all(simg[n:] == -1)
    # full ordering of the first 60 tuples (budget = infinity -> everything valid is taken, in order)
    n2, simg2, sid2, _ = _device_select(ops, scores, valid, paths, None, 10 ** 9)
    assert n2 == int(valid.sum())
    assert np.array_equal(simg2[:60], g['sorted_img']) and np.array_equal(sid2[:60], g['sorted_id'])


@pytest.mark.parametrize("n_img,S,budget,fair", [(5, 16, 7, True), (40, 150, 300, True), (64, 2048, 5000, False),
                                                  (3, 7, 1000, True), (9, 33, 0, False)])
def test_tie_heavy_random_pools(n_img, S, budget, fair):
    """Scores quantised to a handful of values -> almost every comparison is decided by the path-string
    and id tie-breaks; paths are deliberately NOT in index order."""
    ops = _gpu()
    rs = np.random.RandomState(n_img * 1000 + S)
    scores = (rs.randint(0, 6, size=(n_img, S)) / 5.0).astype(np.float32)
    scores[rs.uniform(size=scores.shape) < 0.1] = 0.0
    valid = (rs.uniform(size=(n_img, S)) < 0.8).astype(np.uint8)
    names = rs.permutation(n_img)
    paths = ["a/img_%d.png,b/lbl_%d.png,c/spx_%d.pkl" % (k, k, k) for k in names]   # "img_10" < "img_9" etc.
    cost = rs.randint(1, 5, size=(n_img, S))
    bits = cost
    ref = _python_select(scores, valid, paths, cost if fair else None, budget)
    n, simg, sid, ssc = _device_select(ops, scores, valid, paths, bits if fair else None, budget, max_out=budget + 1)
    assert n == len(ref)
    assert [(int(a), int(b)) for a, b in zip(simg[:n], sid[:n])] == [(a, b) for a, b, _ in ref]
    assert np.array_equal(ssc[:n], np.array([c for _, _, c in ref], dtype=np.float32))


def test_negative_and_zero_scores_order():
    """my_bvsb's min-max normalisation can emit zeros and (for absent regions) negatives."""
    ops = _gpu()
    scores = np.array([[0.5, -0.25, 0.0, 1.0], [-1.0, 0.0, 0.75, -0.25]], dtype=np.float32)
    paths = ["p0", "p1"]
    ref = _python_select(scores, None, paths, None, 100)
    n, simg, sid, _ = _device_select(ops, scores, None, paths, None, 100)
    assert n == 8
    assert [(int(a), int(b)) for a, b in zip(simg[:n], sid[:n])] == [(a, b) for a, b, _ in ref]


def test_minmax_normalize_matches_plain_bvsb_reference():
    """my_bvsb.py:79-81 on the golden's un-normalised region means is not stored; check the formula
    against numpy f32 arithmetic instead (sub and div are correctly rounded on both sides)."""
    ops = _gpu()
    rs = np.random.RandomState(3)
    u = rs.uniform(1e-8, 0.9, size=(7, 150)).astype(np.float32)
    u[rs.uniform(size=u.shape) < 0.2] = 0.0
    mn = u[u != 0].min()
    ref = (u - mn)
    ref = ref / ref.max()
    out = ops.minmax_normalize_(torch.from_numpy(u.copy()).cuda()).cpu().numpy()
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_ordering_of_per_rank_heads_equals_the_replicated_ordering(world):
    """engine.select_regions on several ranks (HipBackend.local_head per rank + walk_heads on the gathered heads, here with the
    ranks simulated one after the other on one GPU) consumes exactly the prefix backend.select consumes on all regions: ties
    across ranks, invalid regions, a ragged last shard, fair-counting costs."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.active_selection.engine import HipBackend, ShardPlan
    dev = torch.device('cuda:0')
    be = HipBackend(dev)
    n_img, S, bs = 37, 96, 4
    g = torch.Generator(device=dev).manual_seed(world)
    scores = (torch.randint(0, 50, (n_img, S), generator=g, device=dev).float() / 50.0).contiguous()      # heavy ties
    valid = (torch.rand((n_img, S), generator=g, device=dev) > 0.1).to(torch.uint8)
    cost = (torch.randint(1, 5, (n_img, S), generator=g, device=dev)).to(torch.uint8)
    perm = torch.randperm(n_img, generator=g, device=dev).to(torch.int32)                                   # path ranks != picture order
    inv = torch.empty_like(perm)
    inv[perm.long()] = torch.arange(n_img, dtype=torch.int32, device=dev)
    budget = 400
    want = be.select(scores, valid, perm, inv, cost, budget, budget + 1)
    heads = torch.cat([be.local_head(ShardPlan(n_img, bs, r, world), scores, valid, perm, budget + 1) for r in range(world)])
    got = be.walk_heads(heads, cost, inv, S, budget, budget + 1)
    assert got[0] == want[0] and got[0] > 50
    for a, b in zip(got[1:], want[1:]):
        assert np.array_equal(a, b)
