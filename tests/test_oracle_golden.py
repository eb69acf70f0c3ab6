"""Pin the CPU restatement (oracle/port.py) to the golden vectors produced by running the
reference's own Python (oracle/gen_golden.py).  CPU-only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from mulactseg_amd import synth
from oracle import port

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def single_thread():
    """The goldens were produced with torch.set_num_threads(1): ATen's CPU kernels change their f32
    rounding by 1 ulp with the thread count (chunked vectorisation), so bit-equality with the executed
    reference is only defined at a fixed thread count."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def pool_inputs(seed, n_img, C, H, W, S, n_removed):
    z = synth.logits(seed, n_img, C, H, W)
    spx = np.stack([synth.superpixel_map(seed * 31 + i, H, W, S, n_missing=(1 if i == 1 else 0))
                    for i in range(n_img)])
    im_idx = [["leftImg8bit/train/c/img_%04d.png" % i, "gtFine/train/c/lbl_%04d.png" % i,
               "superpixel/train/c/spx_%04d.pkl" % i] for i in range(n_img)]
    rs = np.random.RandomState(seed + 5)
    suppix = {}
    for i in range(n_img):
        present = sorted(set(np.unique(spx[i]).tolist()))
        removed = set(rs.choice(present, size=n_removed, replace=False).tolist())
        suppix[im_idx[i][2]] = [s for s in present if s not in removed]
    return z, spx, im_idx, suppix


def g1_inputs(g):
    C = int(g['C'])
    z, spx, im_idx, suppix = pool_inputs(int(g['seed']), int(g['n_img']), C, int(g['H']), int(g['W']),
                                         int(g['S']), int(g['n_removed']))
    z[0, C - 1, :16, :24] += 1.5
    assert digest(z, spx) == g['input_digest']
    return z, spx, im_idx, suppix


def g2_inputs(g):
    z, spx, im_idx, suppix = pool_inputs(int(g['seed']), int(g['n_img']), int(g['C']), int(g['H']),
                                         int(g['W']), int(g['S']), int(g['n_removed']))
    assert digest(z, spx) == g['input_digest']
    return z, spx, im_idx, suppix


def loss_inputs(seed, N, C, H, W, S):
    z = synth.logits(seed, N, C, H, W)
    spx, msk = [], []
    for i in range(N):
        s, m = synth.train_crop(seed * 17 + i, H, W, S, frac_selected=0.25)
        spx.append(s)
        msk.append(m)
    spx, msk = np.stack(spx), np.stack(msk)
    tgt = np.stack([synth.multi_hot_targets(seed * 19 + i, S, C) for i in range(N)])
    msk[2] = False
    onehot = tgt[3].sum(axis=1) == 1
    msk[3] &= np.concatenate([onehot, [False]])[spx[3]]
    return z, tgt, spx, msk


def g3_inputs(g):
    z, tgt, spx, msk = loss_inputs(int(g['seed']), int(g['N']), int(g['C']), int(g['H']), int(g['W']),
                                   int(g['S']))
    assert digest(z, tgt, spx, msk) == g['input_digest']
    return z, tgt, spx, msk


def tuples_to_arrays(tuples, im_idx):
    paths = {','.join(k): n for n, k in enumerate(im_idx)}
    return (np.array([t[0] for t in tuples], dtype=np.float64),
            np.array([paths[t[1]] for t in tuples], dtype=np.int64),
            np.array([t[2] for t in tuples], dtype=np.int64))


def test_g1_pixbal_city_port_matches_reference():
    g = np.load(os.path.join(GOLDEN, "g1_pixbal_city.npz"))
    z, spx, im_idx, suppix = g1_inputs(g)
    r = port.pixbal_scores(torch.from_numpy(z), torch.from_numpy(spx), int(g['batch_size']),
                           float(g['ce_temp']), float(g['coeff']), int(g['S']), ban_ignore=True)
    # same torch ops in the same order -> bit-identical to the executed reference
    assert np.array_equal(r['cum'].numpy(), g['cum'])
    assert np.array_equal(r['cls_weight'].numpy(), g['cls_weight'])
    assert np.array_equal(r['region_ntop1'].numpy(), g['region_ntop1'])
    assert np.array_equal(r['scores'].numpy(), g['scores_tensor'])
    tuples = port.score_list(im_idx, suppix, r['scores'])
    sc, si, sid = tuples_to_arrays(tuples, im_idx)
    assert np.array_equal(sc, g['list_score']) and np.array_equal(si, g['list_img']) \
        and np.array_equal(sid, g['list_id'])
    mh = g['multi_hot']
    idx_of = {','.join(k): n for n, k in enumerate(im_idx)}
    consumed = port.select_regions(tuples, int(g['budget']),
                                   cost_fn=lambda path, rid: int(mh[idx_of[path], rid].sum()))
    cc, ci, cid = tuples_to_arrays(consumed, im_idx)
    assert np.array_equal(cc, g['consumed_score']) and np.array_equal(ci, g['consumed_img']) \
        and np.array_equal(cid, g['consumed_id'])
    oc, oi, oid = tuples_to_arrays(sorted(tuples, reverse=True)[:60], im_idx)
    assert np.array_equal(oi, g['sorted_img']) and np.array_equal(oid, g['sorted_id'])
    sel = np.zeros_like(g['isselected'])
    sel[ci, cid] = 1
    assert np.array_equal(sel, g['isselected'])


def test_g2_voc_port_matches_reference():
    g = np.load(os.path.join(GOLDEN, "g2_voc.npz"))
    z, spx, im_idx, suppix = g2_inputs(g)
    zt, st = torch.from_numpy(z), torch.from_numpy(spx)
    S, bs = int(g['S']), int(g['batch_size'])
    for tag, strip in (('plain', False), ('strip', True)):
        s = port.bvsb_scores(zt, st, bs, float(g['ce_temp']), S, strip_last=strip)
        assert np.array_equal(s.numpy(), g['bvsb_%s_scores_tensor' % tag])
        sc, si, sid = tuples_to_arrays(port.score_list(im_idx, suppix, s), im_idx)
        assert np.array_equal(sc, g['bvsb_%s_list_score' % tag])
        assert np.array_equal(sid, g['bvsb_%s_list_id' % tag])
    r = port.pixbal_scores(zt, st, bs, float(g['ce_temp']), float(g['coeff']), S, ban_ignore=False)
    assert np.array_equal(r['cum'].numpy(), g['pwr_cum'])
    assert np.array_equal(r['cls_weight'].numpy(), g['pwr_cls_weight'])
    assert np.array_equal(r['region_ntop1'].numpy(), g['pwr_region_ntop1'])
    assert np.array_equal(r['scores'].numpy(), g['pwr_scores_tensor'])


LOSS_CASES = [
    ('decomp', lambda z, t, s, m, S, T: port.merged_positive_ce(z, t, s, m, T, 'decomp')),
    ('onlymulti', lambda z, t, s, m, S, T: port.group_max_ce(z, t, s, m, S, T, 'onlymulti')),
    ('mc_predignore', lambda z, t, s, m, S, T: port.merged_positive_ce(z, t, s, m, T, 'predignore')),
    ('group_predignore', lambda z, t, s, m, S, T: port.group_max_ce(z, t, s, m, S, T, 'predignore')),
]


@pytest.mark.parametrize("tag,fn", LOSS_CASES, ids=[c[0] for c in LOSS_CASES])
def test_g3_losses_port_matches_reference(tag, fn):
    g = np.load(os.path.join(GOLDEN, "g3_losses.npz"))
    z, tgt, spx, msk = g3_inputs(g)
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    res = fn(zt, torch.from_numpy(tgt), torch.from_numpy(spx), torch.from_numpy(msk), int(g['S']),
             float(g['temp']))
    res = res if isinstance(res, tuple) else (res,)
    for k, r in enumerate(res):
        assert np.float32(float(r.detach())) == g['%s_loss%d' % (tag, k)]
        (gr,) = torch.autograd.grad(r, zt, retain_graph=True)
        assert np.array_equal(gr.numpy(), g['%s_grad%d' % (tag, k)])


def test_g3_base_variants_and_total():
    g = np.load(os.path.join(GOLDEN, "g3_losses.npz"))
    z, tgt, spx, msk = g3_inputs(g)
    N, S, T = int(g['N']), int(g['S']), float(g['temp'])
    tb = torch.from_numpy(np.concatenate([tgt, np.zeros((N, S, 1), np.uint8)], axis=2))
    ts, tm = torch.from_numpy(spx), torch.from_numpy(msk)
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    l = port.merged_positive_ce(zt, tb, ts, tm, T, 'base')
    assert np.float32(float(l)) == g['mc_base_loss0']
    assert np.array_equal(torch.autograd.grad(l, zt)[0].numpy(), g['mc_base_grad0'])
    l = port.group_max_ce(zt, tb, ts, tm, S, T, 'base')
    assert np.float32(float(l)) == g['group_base_loss0']
    assert np.array_equal(torch.autograd.grad(l, zt)[0].numpy(), g['group_base_grad0'])
    # production combination (lossdecomp.py:102-104)
    tt = torch.from_numpy(tgt)
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    group = port.group_max_ce(zt, tt, ts, tm, S, T, 'onlymulti')
    ce, mc = port.merged_positive_ce(zt, tt, ts, tm, T, 'decomp')
    total = 16.0 * ce + 8.0 * mc + 1.0 * group
    total.backward()
    assert np.float32(float(total)) == g['total_loss']
    assert np.array_equal(zt.grad.numpy(), g['total_grad'])
    # stage-2 temperature CE
    rs = np.random.RandomState(int(g['seed']) + 3)
    y = rs.randint(0, int(g['C']), size=(N, int(g['H']), int(g['W']))).astype(np.int64)
    y[rs.uniform(size=y.shape) < 0.2] = 255
    assert digest(y) == g['tce_digest']
    zt = torch.from_numpy(z).clone().requires_grad_(True)
    l2 = port.temperature_ce(zt, torch.from_numpy(y), T)
    l2.backward()
    assert np.float32(float(l2)) == g['tce_loss']
    assert np.array_equal(zt.grad.numpy(), g['tce_grad'])


def test_g5_miou_port_matches_reference():
    g = np.load(os.path.join(GOLDEN, "g5_miou.npz"))
    rs = np.random.RandomState(int(g['seed']))
    B, H, W, nc = int(g['B']), int(g['H']), int(g['W']), int(g['nc'])
    logits = rs.standard_normal(size=(2, B, nc + 1, H, W)).astype(np.float32)
    labels = rs.randint(0, nc, size=(2, B, H, W)).astype(np.int64)
    labels[labels == 7] = 3
    labels[rs.uniform(size=labels.shape) < 0.15] = 255
    assert digest(logits, labels) == g['input_digest']
    seen = np.zeros(nc); correct = np.zeros(nc); positive = np.zeros(nc)
    ign = np.zeros(3)
    for step in range(2):
        p = torch.from_numpy(logits[step]); t = torch.from_numpy(labels[step])
        s, c, q = port.iou_counts(p[:, :-1].max(dim=1)[1], t, nc, 255)
        seen += s; correct += c; positive += q
        ign += np.array(port.ignore_iou_counts(p.max(dim=1)[1], t, nc, 255))
    assert np.array_equal(seen, g['seen']) and np.array_equal(correct, g['correct']) \
        and np.array_equal(positive, g['positive'])
    ious = port.ious_from_counts(seen, correct, positive)
    assert np.array_equal(np.array(ious), g['ious'])
    assert ious[7] == 100
    assert np.mean(ious) == g['miou']
    assert np.array_equal(ign, g['ign'])


G7 = (('banignore', True, False), ('clsbal_banignore', True, True), ('clsbal', False, True))


def g7_inputs(g, tag):
    z, spx, im_idx, suppix = pool_inputs(int(g[tag + '_seed']), int(g['n_img']), int(g[tag + '_C']), int(g['H']), int(g['W']),
                                         int(g['S']), 4)
    if int(g[tag + '_C']) == 20:
        z[0, 19, :14, :20] += 1.5
    assert digest(z, spx) == g[tag + '_digest']
    return z, spx, im_idx, suppix


@pytest.mark.parametrize("tag,ban,bal", G7, ids=[t[0] for t in G7])
def test_g7_remaining_selectors_port_matches_reference(tag, ban, bal):
    g = np.load(os.path.join(GOLDEN, "g7_selectors.npz"))
    z, spx, im_idx, suppix = g7_inputs(g, tag)
    s, w = port.bvsb_variant_scores(torch.from_numpy(z), torch.from_numpy(spx), int(g['batch_size']), 0.1, int(g['S']), ban, bal)
    assert np.array_equal(s.numpy(), g[tag + '_scores_tensor'])
    if bal:
        assert np.array_equal(w.numpy(), g[tag + '_cls_weight'])
    sc, si, sid = tuples_to_arrays(port.score_list(im_idx, suppix, s), im_idx)
    assert np.array_equal(sc, g[tag + '_list_score']) and np.array_equal(sid, g[tag + '_list_id'])


def stage2_inputs(seed, N, C, Ch, H, W, S):
    rs = np.random.RandomState(seed)
    cm = np.stack([synth.class_map(seed * 3 + i, H, W, C, blob=10) for i in range(N)])
    proto = rs.standard_normal((C, Ch)).astype(np.float32)
    feats = proto[cm].transpose(0, 3, 1, 2) + 0.6 * rs.standard_normal((N, Ch, H, W)).astype(np.float32)
    feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)
    z = synth.logits(seed + 1, N, C, H, W)
    spx = np.stack([synth.superpixel_map(seed * 5 + i, H, W, S) for i in range(N)])
    tgt = np.stack([synth.multi_hot_targets(seed * 7 + i, S, C, p_counts=(0.5, 0.3, 0.15, 0.05)) for i in range(N)])
    msk = np.zeros((N, H, W), dtype=bool)
    for i in range(N):
        chosen = rs.choice(S, size=max(3, S // 6), replace=False)
        msk[i] = np.isin(spx[i], chosen)
    msk[N - 1] = False
    labels = rs.randint(0, C - 1, size=(N, H, W)).astype(np.int64)
    return feats, z, tgt, spx, msk, labels


def g6_inputs(g):
    feats, z, tgt, spx, msk, labels = stage2_inputs(int(g['seed']), int(g['N']), int(g['C']), int(g['Ch']), int(g['H']),
                                                    int(g['W']), int(g['S']))
    assert digest(feats, z, tgt, spx, msk) == g['input_digest']
    return feats, z, tgt, spx, msk


@pytest.mark.parametrize("tag,include", [('multi', False), ('all', True)])
def test_g6_stage2_port_matches_reference(tag, include):
    g = np.load(os.path.join(GOLDEN, "g6_stage2.npz"))
    feats, z, tgt, spx, msk = g6_inputs(g)
    out = port.cosine_pseudo_labels(torch.from_numpy(feats), torch.from_numpy(z), torch.from_numpy(tgt), torch.from_numpy(msk),
                                    torch.from_numpy(spx), int(g['S']), include)
    assert np.array_equal(out.numpy().astype(np.int16), g['plbl_' + tag])
    assert int((out[-1] != 255).sum()) == 0              # nothing selected -> nothing labelled
