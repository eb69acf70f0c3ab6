"""GPU parity of the stage-1 loss kernels through the C ABI:
  * bit-exact against the plain-C restatement (oracle/exact.c) -- sums, counts, arg-pixel table, dz;
  * within 1e-4 (north-star tolerance; observed ~5e-7) of the golden vectors produced by executing the
    reference's loss classes (tests/golden/g3_losses.npz)."""
import os

import numpy as np
import pytest
import torch

from mulactseg_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    return ops


def _inputs(seed, N, C, H, W, S, frac=0.25):
    z = synth.logits(seed, N, C, H, W)
    spx, msk = [], []
    for i in range(N):
        s, m = synth.train_crop(seed * 17 + i, H, W, S, frac_selected=frac)
        spx.append(s)
        msk.append(m)
    tgt = np.stack([synth.multi_hot_targets(seed * 19 + i, S, C) for i in range(N)])
    return z, tgt, np.stack(spx), np.stack(msk)


FLAG_SETS = {
    'production': 1 | 2 | 4 | 8,
    'decomp': 1 | 8,
    'onlymulti': 2 | 4,
    'mc_predignore': 1,
    'group_predignore': 2,
}
SHAPES = [(2, 20, 64, 512, 128), (3, 20, 40, 44, 48), (2, 21, 33, 37, 150), (1, 7, 24, 36, 16), (2, 20, 192, 768, 512),
          # BASELINE.json's literal "769x769x19" crop (the reference's transform is NAMED 769 but crops 768, transform.py:91-114):
          # odd row length -> rows are only 4-byte aligned, every scan takes its per-row alignment prologue
          (1, 20, 769, 769, 2048), (1, 19, 769, 769, 2048)]


@pytest.mark.parametrize("N,C,H,W,S", SHAPES)
@pytest.mark.parametrize("fname", sorted(FLAG_SETS))
def test_losses_bit_exact_vs_c_oracle(N, C, H, W, S, fname):
    ops = _gpu()
    from oracle import exact
    flags = FLAG_SETS[fname]
    z, tgt, spx, msk = _inputs(300 + W + C, N, C, H, W, S)
    invT = ops.inv_temperature(0.1)
    ebits = exact.target_bits(tgt)
    eacc, egmax, eloss = exact.partial_loss_fwd(z, spx, msk, ebits, np.float32(invT), flags)
    zt, st, mt = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    bits = ops.target_bits(torch.from_numpy(tgt).cuda())
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), ebits)
    losses, acc, gmax = ops.partial_loss_fwd(zt, st, mt, bits, invT, flags)
    assert np.array_equal(acc.cpu().numpy().view(np.uint64), eacc)
    if gmax is not None:
        assert np.array_equal(gmax.cpu().numpy().view(np.uint64), egmax)
    assert np.array_equal(losses.cpu().numpy(), eloss)
    go = np.array([16.0, 8.0, 1.0], dtype=np.float32)
    _, edz = exact.partial_loss_bwd(z, spx, msk, ebits, egmax, eacc, go, np.float32(invT), flags)
    dz = ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, torch.from_numpy(go).cuda(), invT, flags)
    assert np.array_equal(dz.cpu().numpy(), edz)
    # int32 ids give the same bits
    losses2, acc2, _ = ops.partial_loss_fwd(zt, st.to(torch.int32), mt, bits, invT, flags)
    assert torch.equal(acc2, acc)


def _g3():
    g = np.load(os.path.join(GOLDEN, "g3_losses.npz"))
    z, tgt, spx, msk = _inputs(int(g['seed']), int(g['N']), int(g['C']), int(g['H']), int(g['W']), int(g['S']))
    msk[2] = False
    onehot = tgt[3].sum(axis=1) == 1
    msk[3] &= np.concatenate([onehot, [False]])[spx[3]]
    return g, z, tgt, spx, msk


def _close(a, b, tol=1e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))


def test_modules_match_reference_golden():
    """The nn.Module surface (same names / signatures as the reference) against the executed reference."""
    _gpu()
    from mulactseg_amd.utils import loss as L
    g, z, tgt, spx, msk = _g3()
    T, S, C, N = float(g['temp']), int(g['S']), int(g['C']), int(g['N'])
    tt, ts, tm = torch.from_numpy(tgt).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    tb = torch.from_numpy(np.concatenate([tgt, np.zeros((N, S, 1), np.uint8)], axis=2)).cuda()

    def run(fn, tag, n_out):
        zt = torch.from_numpy(z).cuda().requires_grad_(True)
        res = fn(zt)
        res = res if isinstance(res, tuple) else (res,)
        assert len(res) == n_out
        for k, r in enumerate(res):
            assert abs(float(r) - float(g['%s_loss%d' % (tag, k)])) <= 1e-4 * max(1.0, abs(float(g['%s_loss%d' % (tag, k)])))
            (gr,) = torch.autograd.grad(r, zt, retain_graph=True)
            ref = g['%s_grad%d' % (tag, k)]
            assert np.max(np.abs(gr.cpu().numpy() - ref)) <= 1e-4 * np.max(np.abs(ref)) + 1e-9

    run(lambda zt: L.OnehotCEMultihotChoice(num_class=C - 1, temperature=T)(zt, tt, ts, tm), 'decomp', 2)
    run(lambda zt: L.GroupMultiLabelCE_onlymulti(args=None, num_class=C - 1, num_superpixel=S, temperature=T)(zt, tt, ts, tm), 'onlymulti', 1)
    run(lambda zt: L.MultiChoiceCE_(num_class=C - 1, temperature=T)(zt, tt, ts, tm), 'mc_predignore', 1)
    run(lambda zt: L.GroupMultiLabelCE_(args=None, num_class=C - 1, num_superpixel=S, temperature=T)(zt, tt, ts, tm), 'group_predignore', 1)
    run(lambda zt: L.MultiChoiceCE(num_class=C, temperature=T)(zt, tb, ts, tm), 'mc_base', 1)
    run(lambda zt: L.GroupMultiLabelCE(args=None, num_class=C, num_superpixel=S, temperature=T)(zt, tb, ts, tm), 'group_base', 1)
    # production combination through the fused module (train_impl: 16*ce + 8*mc + 1*group)
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    group, ce, mc = L.FusedPartialLabelLoss(S, T, T)(zt, tt, ts, tm)
    total = 16.0 * ce + 8.0 * mc + 1.0 * group
    total.backward()
    assert abs(float(total) - float(g['total_loss'])) <= 1e-4 * float(g['total_loss'])
    assert np.max(np.abs(zt.grad.cpu().numpy() - g['total_grad'])) <= 1e-4 * np.max(np.abs(g['total_grad']))
    # the same through the two separate modules (drop-in surface) gives the same gradient bits
    zt2 = torch.from_numpy(z).cuda().requires_grad_(True)
    g2 = L.GroupMultiLabelCE_onlymulti(args=None, num_class=C - 1, num_superpixel=S, temperature=T)(zt2, tt, ts, tm)
    ce2, mc2 = L.OnehotCEMultihotChoice(num_class=C - 1, temperature=T)(zt2, tt, ts, tm)
    assert float(ce2) == float(ce) and float(mc2) == float(mc) and float(g2) == float(group)
    # stage-2 temperature CE
    rs = np.random.RandomState(int(g['seed']) + 3)
    y = rs.randint(0, C, size=(N, int(g['H']), int(g['W']))).astype(np.int64)
    y[rs.uniform(size=y.shape) < 0.2] = 255
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    l2 = L.MyCrossEntropyLoss(ignore_index=255, temperature=T)(zt, torch.from_numpy(y).cuda())
    l2.backward()
    assert abs(float(l2) - float(g['tce_loss'])) <= 1e-4 * float(g['tce_loss'])
    assert np.max(np.abs(zt.grad.cpu().numpy() - g['tce_grad'])) <= 1e-4 * np.max(np.abs(g['tce_grad']))


def test_empty_mask_gives_zero_loss_and_zero_grad():
    """loss == 0 path of the reference (update() skips the step, active_joint_multi.py:23-37)."""
    _gpu()
    from mulactseg_amd.utils import loss as L
    N, C, H, W, S = 2, 20, 32, 256, 16
    z, tgt, spx, msk = _inputs(5, N, C, H, W, S)
    msk[:] = False
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    group, ce, mc = L.FusedPartialLabelLoss(S, 0.1, 0.1)(zt, torch.from_numpy(tgt).cuda(), torch.from_numpy(spx).cuda(),
                                                         torch.from_numpy(msk).cuda())
    total = 16 * ce + 8 * mc + group
    assert float(total) == 0.0
    total.backward()
    assert float(zt.grad.abs().max()) == 0.0


def test_gradient_only_on_selected_pixels_and_linear_in_upstream():
    """Size-independent properties at the full training shape [4,20,768,768]: dz is zero off-mask,
    per-pixel gradients sum to ~0 over classes (softmax), and dz is linear in the upstream gradient."""
    ops = _gpu()
    N, C, H, W, S = 4, 20, 768, 768, 2048
    z, tgt, spx, msk = _inputs(11, N, C, H, W, S, frac=0.09)
    zt, st, mt = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    bits = ops.target_bits(torch.from_numpy(tgt).cuda())
    invT = ops.inv_temperature(0.1)
    flags = 1 | 2 | 4 | 8
    losses, acc, gmax = ops.partial_loss_fwd(zt, st, mt, bits, invT, flags)
    a = acc.cpu().numpy()
    assert a[2] + a[3] + a[4] == int(msk.sum())           # every selected pixel is counted exactly once
    g1 = ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, torch.tensor([16.0, 8.0, 1.0]).cuda(), invT, flags)
    g2 = ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, torch.tensor([32.0, 16.0, 2.0]).cuda(), invT, flags)
    assert float((g1 * (~mt).unsqueeze(1)).abs().max()) == 0.0
    assert torch.equal(g2, 2 * g1)                         # exact: power-of-two scaling
    assert float(g1.sum(dim=1).abs().max()) < 1e-5 * float(g1.abs().max()) + 1e-7
    # group-loss table: one arg pixel per (superpixel, class) entry, inside the mask, in that superpixel
    gm = gmax.cpu().numpy().view(np.uint64)
    nz = np.argwhere(gm != 0)
    pix = (0xffffffff - (gm[gm != 0] & np.uint64(0xffffffff))).astype(np.int64)
    assert np.all(msk.reshape(N, -1)[nz[:, 0], pix])
    assert np.all(spx.reshape(N, -1)[nz[:, 0], pix] == nz[:, 1])
    assert int(a[6]) == len(nz)


def test_full_training_batch_properties():
    """BASELINE configs[1] shape [4,20,768,768], S = 2048: one image bit-exact against the C oracle; over the batch the
    integer accumulators are additive over images, the gradient is zero outside the selection, and permuting the
    images permutes dz (no cross-image leakage); selected fractions 0 %, ~9 % and 100 % in the same batch."""
    ops = _gpu()
    from oracle import exact
    N, C, H, W, S = 4, 20, 768, 768, 2048
    flags = FLAG_SETS['production']
    z, tgt, spx, msk = _inputs(777, N, C, H, W, S, frac=0.09)
    msk[1] = False                                   # nothing selected
    msk[2] = spx[2] < S                              # everything but the pad id selected
    invT = ops.inv_temperature(0.1)
    zt, st, mt = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    bits = ops.target_bits(torch.from_numpy(tgt).cuda())
    losses, acc, gmax = ops.partial_loss_fwd(zt, st, mt, bits, invT, flags)
    go = torch.tensor([16.0, 8.0, 1.0], device='cuda')
    dz = ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, go, invT, flags)
    assert bool(torch.isfinite(dz).all()) and float(dz[1].abs().max()) == 0.0
    assert float((dz * (~mt)[:, None]).abs().max()) == 0.0
    # additivity of the integer sums / counts over images
    tot = torch.zeros_like(acc)
    for i in range(N):
        _, a_i, g_i = ops.partial_loss_fwd(zt[i:i + 1], st[i:i + 1], mt[i:i + 1], bits[i:i + 1], invT, flags)
        tot += a_i
        assert torch.equal(g_i[0], gmax[i])
    assert torch.equal(tot, acc)
    # image 0 against the oracle, bit for bit (forward tables and sums)
    ebits = exact.target_bits(tgt[:1])
    eacc, egmax, _ = exact.partial_loss_fwd(z[:1], spx[:1], msk[:1], ebits, np.float32(invT), flags)
    _, a0, g0 = ops.partial_loss_fwd(zt[:1], st[:1], mt[:1], bits[:1], invT, flags)
    assert np.array_equal(a0.cpu().numpy().view(np.uint64), eacc) and np.array_equal(g0.cpu().numpy().view(np.uint64), egmax)
    # ... and image 0's gradient: the scales come from the batch-wide accumulators (normalisers 1 + n over the batch)
    _, edz0 = exact.partial_loss_bwd(z[:1], spx[:1], msk[:1], ebits, egmax, acc.cpu().numpy().view(np.uint64),
                                     np.array([16.0, 8.0, 1.0], dtype=np.float32), np.float32(invT), flags)
    assert np.array_equal(dz[0].cpu().numpy(), edz0[0])
    # permutation
    perm = torch.tensor([3, 0, 2, 1], device='cuda')
    l2, acc2, gmax2 = ops.partial_loss_fwd(zt[perm].contiguous(), st[perm].contiguous(), mt[perm].contiguous(), bits[perm].contiguous(), invT, flags)
    assert torch.equal(acc2, acc) and torch.equal(l2, losses)
    dz2 = ops.partial_loss_bwd(zt[perm].contiguous(), st[perm].contiguous(), mt[perm].contiguous(), bits[perm].contiguous(), gmax2, acc2, go, invT, flags)
    assert torch.equal(dz2, dz[perm])


def test_literal_769_crop_batch_properties():
    """[4,20,769,769] (BASELINE.json metric shape): additivity over images, zero gradient off the selection, exact
    power-of-two linearity, and image 3 bit-exact against the C oracle (forward tables, sums and dz)."""
    ops = _gpu()
    from oracle import exact
    N, C, H, W, S = 4, 20, 769, 769, 2048
    flags = FLAG_SETS['production']
    z, tgt, spx, msk = _inputs(769, N, C, H, W, S, frac=0.09)
    invT = ops.inv_temperature(0.1)
    zt, st, mt = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    bits = ops.target_bits(torch.from_numpy(tgt).cuda())
    losses, acc, gmax = ops.partial_loss_fwd(zt, st, mt, bits, invT, flags)
    go = torch.tensor([16.0, 8.0, 1.0], device='cuda')
    dz = ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, go, invT, flags)
    assert float((dz * (~mt)[:, None]).abs().max()) == 0.0
    assert torch.equal(ops.partial_loss_bwd(zt, st, mt, bits, gmax, acc, 4 * go, invT, flags), 4 * dz)
    tot = torch.zeros_like(acc)
    for i in range(N):
        tot += ops.partial_loss_fwd(zt[i:i + 1], st[i:i + 1], mt[i:i + 1], bits[i:i + 1], invT, flags)[1]
    assert torch.equal(tot, acc)
    ebits = exact.target_bits(tgt[3:])
    eacc, egmax, _ = exact.partial_loss_fwd(z[3:], spx[3:], msk[3:], ebits, np.float32(invT), flags)
    _, a3, g3 = ops.partial_loss_fwd(zt[3:], st[3:], mt[3:], bits[3:], invT, flags)
    assert np.array_equal(a3.cpu().numpy().view(np.uint64), eacc) and np.array_equal(g3.cpu().numpy().view(np.uint64), egmax)
    _, edz = exact.partial_loss_bwd(z[3:], spx[3:], msk[3:], ebits, egmax, acc.cpu().numpy().view(np.uint64),
                                    np.array([16.0, 8.0, 1.0], dtype=np.float32), np.float32(invT), flags)
    assert np.array_equal(dz[3].cpu().numpy(), edz[0])


# ---------------------------------------------------------------------------------------------------------------------
# quarter-resolution forms: the x4 bilinear upsampling of the logits (models/segmentation/utils.py:25) inside the scans
# ---------------------------------------------------------------------------------------------------------------------
LOW_SHAPES = [(2, 20, 16, 48, 64, 192, 128), (1, 20, 13, 11, 52, 44, 48), (2, 21, 9, 10, 33, 37, 150), (1, 20, 48, 48, 192, 192, 512),
              (1, 20, 193, 193, 769, 769, 2048)]


@pytest.mark.parametrize("N,C,h,w,H,W,S", LOW_SHAPES)
@pytest.mark.parametrize("fname", ['production', 'decomp', 'group_predignore'])
def test_lowres_scans_equal_the_materialised_path_and_the_c_oracle(N, C, h, w, H, W, S, fname):
    """Forward: sums, counts, arg-pixel table and losses are BIT-identical to the full-resolution scan of
    upsample_bilinear(zq) (and the upsampling itself to oracle/exact.c).  Backward: the fixed-point gradient sums of zq equal
    oracle/exact.c:exact_partial_loss_bwd_lowres bit for bit, and the f32 gradient agrees with the composed float path
    (full-resolution dz gathered by the deterministic upsample backward) to rounding."""
    ops = _gpu()
    from oracle import exact
    flags = FLAG_SETS[fname]
    _, tgt, spx, msk = _inputs(900 + W + C, N, C, H, W, S)
    zq = synth.logits(77 + h, N, C, h, w)
    invT = ops.inv_temperature(0.1)
    zqt, st, mt = torch.from_numpy(zq).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    bits = ops.target_bits(torch.from_numpy(tgt).cuda())
    zfull = ops.upsample_bilinear(zqt, (H, W))
    assert np.array_equal(zfull.cpu().numpy(), exact.upsample_bilinear(zq, H, W))
    l_full, a_full, g_full = ops.partial_loss_fwd(zfull, st, mt, bits, invT, flags)
    l_low, a_low, g_low = ops.partial_loss_fwd_lowres(zqt, (H, W), st, mt, bits, invT, flags)
    assert torch.equal(a_low, a_full) and torch.equal(l_low.view(torch.int32), l_full.view(torch.int32))
    if g_full is not None:
        assert torch.equal(g_low, g_full)
    go = np.array([16.0, 8.0, 1.0], dtype=np.float32)
    got = torch.from_numpy(go).cuda()
    dzq, fix = ops.partial_loss_bwd_lowres(zqt, (H, W), st, mt, bits, g_low, a_low, got, invT, flags, want_fix=True)
    egmax = np.zeros((N, S, C), dtype=np.uint64) if g_low is None else g_low.cpu().numpy().view(np.uint64)
    efix, edzq = exact.partial_loss_bwd_lowres(zq, H, W, spx, msk, exact.target_bits(tgt), egmax, a_low.cpu().numpy().view(np.uint64),
                                               go, np.float32(invT), flags)
    assert np.array_equal(fix.cpu().numpy(), efix)
    assert np.array_equal(dzq.cpu().numpy().view(np.uint32), edzq.view(np.uint32))
    # composed float path: dz at full resolution, then the gather backward of the upsampling
    dz = ops.partial_loss_bwd(zfull, st, mt, bits, g_full, a_full, got, invT, flags)
    lib = __import__('mulactseg_amd._lib', fromlist=['x'])
    ref = torch.empty_like(zqt)
    lib.check(lib.load().mas_upsample_bilinear_bwd(dz.data_ptr(), N * C, h, w, H, W, ref.data_ptr(), torch.cuda.current_stream().cuda_stream), "bwd")
    scale = float(ref.abs().max())
    assert scale > 0 and float((dzq - ref).abs().max()) <= 2e-6 * scale
    # run-to-run identical (integer atomics)
    dzq2 = ops.partial_loss_bwd_lowres(zqt, (H, W), st, mt, bits, g_low, a_low, got, invT, flags)
    assert torch.equal(dzq, dzq2)


def test_fused_module_lowres_matches_interpolate_then_loss():
    """FusedPartialLabelLoss.forward_lowres through autograd == F.interpolate + forward: same losses bit for bit, gradient of the
    quarter-resolution logits within rounding (ATen's interpolate backward sums the same terms with float atomics)."""
    _gpu()
    import torch.nn.functional as F
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    N, C, h, w, S = 2, 20, 24, 24, 256
    H, W = 4 * h, 4 * w
    _, tgt, spx, msk = _inputs(41, N, C, H, W, S)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=False)
    tg, sp, mk = torch.from_numpy(tgt).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    zq1 = torch.from_numpy(synth.logits(5, N, C, h, w)).cuda().requires_grad_(True)
    zq2 = zq1.detach().clone().requires_grad_(True)
    g1, c1, m1 = crit.forward_lowres(zq1, (H, W), tg, sp, mk)
    (16.0 * c1 + 8.0 * m1 + g1).backward()
    from mulactseg_amd import ops
    g2, c2, m2 = crit(ops.upsample_bilinear(zq2, (H, W)), tg, sp, mk)
    (16.0 * c2 + 8.0 * m2 + g2).backward()
    assert float(g1) == float(g2) and float(c1) == float(c2) and float(m1) == float(m2)
    s = float(zq2.grad.abs().max())
    assert s > 0 and float((zq1.grad - zq2.grad).abs().max()) <= 2e-6 * s
    ref = F.interpolate(zq2.detach(), size=(H, W), mode='bilinear', align_corners=False)
    assert float((ops.upsample_bilinear(zq2.detach(), (H, W)) - ref).abs().max()) <= 1e-6


def test_weighted_objective_equals_the_torch_composition():
    """FusedPartialLabelLoss.weighted_lowres: total == (16*ce + 8*mc) + 1*group of forward_lowres bit for bit, the parts are the
    same values, and the gradient of the quarter-resolution logits equals the one autograd derives through the torch arithmetic."""
    _gpu()
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    N, C, h, w, S = 2, 20, 24, 40, 256
    H, W = 4 * h, 4 * w
    _, tgt, spx, msk = _inputs(43, N, C, H, W, S)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=False)
    tg, sp, mk = torch.from_numpy(tgt).cuda(), torch.from_numpy(spx).cuda(), torch.from_numpy(msk).cuda()
    z1 = torch.from_numpy(synth.logits(6, N, C, h, w)).cuda().requires_grad_(True)
    z2 = z1.detach().clone().requires_grad_(True)
    total, g1, c1, m1 = crit.weighted_lowres(z1, (H, W), tg, sp, mk, 16.0, 8.0, 1.0)
    total.backward()
    g2, c2, m2 = crit.forward_lowres(z2, (H, W), tg, sp, mk)
    ref = (16.0 * c2) + (8.0 * m2) + (1.0 * g2)
    ref.backward()
    assert float(total) == float(ref) and float(g1) == float(g2) and float(c1) == float(c2) and float(m1) == float(m2)
    assert not g1.requires_grad and total.requires_grad
    assert torch.equal(z1.grad, z2.grad)
    # an upstream factor (the data-parallel trainers multiply the loss by the world size)
    z3 = z1.detach().clone().requires_grad_(True)
    t3, _, _, _ = crit.weighted_lowres(z3, (H, W), tg, sp, mk, 16.0, 8.0, 1.0)
    (t3 * 2.0).backward()
    assert float((z3.grad - 2.0 * z1.grad).abs().max()) <= 1e-6 * float(z1.grad.abs().max())


def test_temperature_ce_kernel_equals_the_c_oracle_and_torch():
    """a-11, ``MyCrossEntropyLoss`` (reference utils/loss.py:10-21) on the MAS_LOSS_TCE form of the scans: value and gradient
    bit for bit equal to oracle/exact.c's twin, within 1e-4 of ATen's CrossEntropyLoss(input / T); the quarter-resolution form
    (``forward_lowres``) equals the materialised form's value bit for bit and ATen's gradient through F.interpolate to 1e-4."""
    ops = _gpu()
    from oracle import exact
    from mulactseg_amd import _lib
    from mulactseg_amd.utils import loss as L
    import torch.nn.functional as F
    N, C, H, W, T = 2, 20, 48, 64, 0.1
    rs = np.random.RandomState(17)
    z = (0.35 * rs.standard_normal((N, C, H, W))).astype(np.float32)
    y = rs.randint(0, C, size=(N, H, W)).astype(np.int64)
    y[rs.uniform(size=y.shape) < 0.25] = 255
    crit = L.MyCrossEntropyLoss(ignore_index=255, temperature=T)
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    yt = torch.from_numpy(y).cuda()
    loss = crit(zt, yt)
    loss.backward()
    # the C twin
    flags = _lib.LOSS_CE | _lib.LOSS_TCE
    invT = np.float32(ops.inv_temperature(T))
    mask = (y != 255)
    bits = np.zeros((N, C), dtype=np.uint32)
    acc, gmax, losses = exact.partial_loss_fwd(z, y, mask, bits, invT, flags)
    assert np.float32(float(loss)) == losses[0]
    _, dz = exact.partial_loss_bwd(z, y, mask, bits, gmax, acc, np.array([1.0, 0.0, 0.0], dtype=np.float32), invT, flags)
    assert np.array_equal(zt.grad.cpu().numpy(), dz)
    # ATen
    zr = torch.from_numpy(z).cuda().requires_grad_(True)
    ref = F.cross_entropy(zr / T, yt, ignore_index=255)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref))
    assert float((zt.grad - zr.grad).abs().max()) <= 1e-4 * float(zr.grad.abs().max())
    # quarter-resolution form
    q = (0.35 * rs.standard_normal((N, C, H // 4, W // 4))).astype(np.float32)
    qt = torch.from_numpy(q).cuda().requires_grad_(True)
    low = crit.forward_lowres(qt, (H, W), yt)
    low.backward()
    up = ops.upsample_bilinear(torch.from_numpy(q).cuda(), (H, W))
    assert float(low) == float(crit(up, yt))
    qr = torch.from_numpy(q).cuda().requires_grad_(True)
    refq = F.cross_entropy(F.interpolate(qr, size=(H, W), mode='bilinear', align_corners=False) / T, yt, ignore_index=255)
    refq.backward()
    assert abs(float(low) - float(refq)) <= 1e-4 * abs(float(refq))
    assert float((qt.grad - qr.grad).abs().max()) <= 1e-4 * float(qr.grad.abs().max())
    # no valid pixel: NaN like torch's mean over nothing
    none = torch.full_like(yt, 255)
    assert torch.isnan(crit(zt.detach(), none))


@pytest.mark.parametrize("fname", sorted(FLAG_SETS))
@pytest.mark.parametrize("low", [False, True])
def test_fused_calls_equal_the_step_by_step_entry_points(fname, low):
    """mas_partial_loss_fwd_fused / _bwd_fused (what the loss modules run: prep + scan + finalize-with-values in one call; the backward
    scan forms its scale factors itself) against the step-by-step entry points the oracle tests above pin: accumulators, arg-pixel
    table, bit masks, loss values, weighted objective and gradients are the same bits -- from u8 target rows or from ready masks,
    with and without weights, and with a reduce hook between scan and division (the data-parallel form)."""
    ops = _gpu()
    flags = FLAG_SETS[fname]
    N, C, S = 2, 20, 200
    h, w, H, W = (24, 36, 96, 144) if low else (0, 0, 40, 100)
    _, tgt, spx, msk = _inputs(1200 + W, N, C, H, W, S)
    z = synth.logits(91, N, C, h if low else H, w if low else W)
    invT = ops.inv_temperature(0.1)
    zt, st, mt, tg = (torch.from_numpy(a).cuda() for a in (z, spx, msk, tgt))
    size = (H, W) if low else None
    bits = ops.target_bits(tg)
    wts = torch.tensor([16.0, 8.0, 1.0], device='cuda')
    if low:
        l0, a0, g0 = ops.partial_loss_fwd_lowres(zt, size, st, mt, bits, invT, flags)
        lw, _, _ = ops.partial_loss_fwd_lowres(zt, size, st, mt, bits, invT, flags, weights=wts)
    else:
        l0, a0, g0 = ops.partial_loss_fwd(zt, st, mt, bits, invT, flags)
        lw = None
    for kw in (dict(targets=tg), dict(bits=bits), dict(targets=tg, reduce_acc=lambda acc: None), dict(targets=tg, weights=wts)):
        losses, state = ops.partial_loss_fwd_fused(zt, size, st, mt, invT, flags, **kw)
        assert torch.equal(state.acc[:7], a0[:7])                 # (word 7 counts the finalize workgroups)
        if g0 is not None:
            assert torch.equal(state.gmax, g0)
        assert torch.equal(losses[:3].view(torch.int32), l0.view(torch.int32))
        if 'weights' in kw and lw is not None:
            assert torch.equal(losses.view(torch.int32), lw.view(torch.int32))
        go3 = torch.tensor([16.0, 8.0, 1.0], device='cuda')
        go1 = torch.tensor([1.0], device='cuda')
        if low:
            ref = ops.partial_loss_bwd_lowres(zt, size, st, mt, bits, g0, a0, go1 if 'weights' in kw else go3, invT, flags,
                                              weights=wts if 'weights' in kw else None)
        else:
            # (full resolution, weighted: the chain rule through the weighted sum is grad * w_k -- the same products)
            ref = ops.partial_loss_bwd(zt, st, mt, bits, g0, a0, go3, invT, flags)
        got = ops.partial_loss_bwd_fused(zt, size, st, mt, state, go1 if 'weights' in kw else go3, invT, weights=wts if 'weights' in kw else None)
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), kw.keys()
    # a second forward on fresh state gives the same values (the prep launch re-zeroes everything, including the finalize counter)
    l2, _ = ops.partial_loss_fwd_fused(zt, size, st, mt, invT, flags, targets=tg)
    assert torch.equal(l2.view(torch.int32), l0.view(torch.int32))
