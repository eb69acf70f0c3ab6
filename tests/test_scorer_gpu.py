"""GPU parity: HIP scorer kernels (through the C ABI) vs the plain-C restatement (oracle/exact.c).
Bit-exact on every output -- integer AND float -- because both sides evaluate the arithmetic of
csrc/detmath.h and accumulate in fixed point."""

import numpy as np
import pytest
import torch

from mulactseg_amd import synth

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    return ops


def _case(seed, B, C, H, W, S):
    z = synth.logits(seed, B, C, H, W)
    spx = np.stack([synth.superpixel_map(seed * 31 + i, H, W, S) for i in range(B)])
    return z, spx


CASES = [
    # B, C, H, W, S
    (2, 20, 64, 512, 128),      # vector path, two tile columns
    (3, 20, 48, 64, 64),        # narrower than a tile
    (2, 21, 33, 37, 150),       # VOC-like, odd sizes -> scalar path
    (1, 19, 40, 260, 32),       # stripped channel count, ragged tile edge
    (2, 7, 24, 36, 16),         # generic (runtime-C) instantiation
    (1, 20, 256, 1024, 2048),   # Cityscapes-like density of regions
    (2, 20, 100, 516, 200),     # ring kernel (H >= 64): ragged last strip (36 rows) and a 4-px-wide last tile column
    (1, 19, 130, 260, 64),      # ring kernel, 19 channels, three strips
    (1, 21, 70, 128, 40),       # ring kernel narrower than a tile
    (2, 7, 96, 36, 16),         # ring kernel, generic (runtime-C) instantiation
    (1, 19, 769, 769, 2048),    # BASELINE.json's literal "769x769x19": odd row length, rows only 4-byte aligned
    (1, 20, 769, 769, 2048),    # the same crop with the "undefined" channel (what the predignore model emits)
]


@pytest.mark.parametrize("B,C,H,W,S", CASES)
@pytest.mark.parametrize("weighted", [False, True])
def test_region_accum_bit_exact(B, C, H, W, S, weighted):
    ops = _gpu()
    from oracle import exact
    z, spx = _case(100 + C + W, B, C, H, W, S)
    invT = ops.inv_temperature(0.1)
    w = None
    if weighted:
        w = (np.random.RandomState(5).uniform(0.2, 1.0, size=C)).astype(np.float32)
    es, eh = exact.bvsb_region_accum(z, spx, w, S, np.float32(invT))
    zt = torch.from_numpy(z).cuda()
    wt = torch.from_numpy(w).cuda() if weighted else None
    for dtype in (torch.int64, torch.int32, torch.int16):
        st = torch.from_numpy(spx).to(dtype).cuda()
        gs, gh = ops.bvsb_region_accum(zt, st, wt, S, invT)
        assert np.array_equal(gs.cpu().numpy().view(np.uint64), es), dtype
        assert np.array_equal(gh.cpu().numpy().view(np.uint32), eh), dtype
    score, dom, cnt, h64 = ops.region_finalize(gs, gh, ban_class=C - 1, want_hist_i64=True)
    escore, edom, ecnt = exact.region_finalize(es, eh, C - 1)
    assert np.array_equal(score.cpu().numpy(), escore)
    assert np.array_equal(dom.cpu().numpy(), edom)
    assert np.array_equal(cnt.cpu().numpy().view(np.uint32), ecnt)
    assert np.array_equal(h64.cpu().numpy(), eh.astype(np.int64))
    assert int(cnt.sum()) == B * H * W


@pytest.mark.parametrize("B,C,H,W,S", CASES)
def test_class_prob_sum_bit_exact(B, C, H, W, S):
    ops = _gpu()
    from oracle import exact
    z, _ = _case(200 + C + W, B, C, H, W, S)
    invT = ops.inv_temperature(0.1)
    e = exact.class_prob_sum(z, np.float32(invT))
    g = ops.class_prob_sum(torch.from_numpy(z).cuda(), invT)
    assert np.array_equal(g.cpu().numpy().view(np.uint64), e)
    # probabilities sum to one: the integer sums add up to about HW * 2^23 per image
    tot = g.cpu().numpy().view(np.uint64).sum(axis=1).astype(np.float64) / 2.0 ** 23 / (H * W)
    assert np.all(np.abs(tot - 1.0) < 1e-5)


def test_out_of_range_ids_are_skipped():
    ops = _gpu()
    from oracle import exact
    B, C, H, W, S = 1, 20, 32, 256, 16
    z, spx = _case(7, B, C, H, W, S)
    spx[:, :, 200:] = S          # pad id, like training crops
    spx[:, 5, :10] = -1
    invT = ops.inv_temperature(0.1)
    es, eh = exact.bvsb_region_accum(z, spx, None, S, np.float32(invT))
    gs, gh = ops.bvsb_region_accum(torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), None, S, invT)
    assert np.array_equal(gs.cpu().numpy().view(np.uint64), es)
    assert np.array_equal(gh.cpu().numpy().view(np.uint32), eh)
    assert int(gh.sum()) == int((spx >= 0).sum() - (spx >= S).sum())


def test_exact_tie_lowest_index_wins():
    ops = _gpu()
    B, C, H, W, S = 1, 20, 8, 64, 4
    z = np.zeros((B, C, H, W), dtype=np.float32)       # every class ties everywhere
    spx = np.zeros((B, H, W), dtype=np.int64)
    gs, gh = ops.bvsb_region_accum(torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda(), None, S,
                                   ops.inv_temperature(0.1))
    h = gh.cpu().numpy()
    assert h[0, 0, 0] == H * W and h.sum() == H * W     # top1 = class 0
    score, dom, cnt, _ = ops.region_finalize(gs, gh)
    assert abs(float(score[0, 0]) - 1.0) < 1e-6         # p2/p1 = 1 (+1e-8)
    assert float(score[0, 1]) == 0.0 and int(dom[0, 1]) == 0


def test_cpu_tensor_is_refused():
    ops = _gpu()
    from mulactseg_amd import _lib
    with pytest.raises(_lib.MulActSegHipError):
        ops.class_prob_sum(torch.zeros(1, 20, 4, 4), 10.0)


@pytest.mark.parametrize("B,C,H,W,S", CASES)
def test_single_pass_bit_exact_and_consistent_with_two_pass(B, C, H, W, S):
    """k_single_pass == oracle bit for bit; its class sums equal the K2 kernel's, its per-class margin sums add up
    to the K3 kernel's unweighted region sums, and with unit weights the finalize equals the two-pass finalize."""
    ops = _gpu()
    from oracle import exact
    z, spx = _case(400 + C + W, B, C, H, W, S)
    invT = ops.inv_temperature(0.1)
    eps_, ecs, eh = exact.single_pass_accum(z, spx, S, np.float32(invT))
    zt = torch.from_numpy(z).cuda()
    for dtype in (torch.int64, torch.int32, torch.int16):
        st = torch.from_numpy(spx).to(dtype).cuda()
        ps, cs, hh = ops.single_pass_accum(zt, st, S, invT)
        assert np.array_equal(ps.cpu().numpy().view(np.uint64), eps_), dtype
        assert np.array_equal(cs.cpu().numpy().view(np.uint64), ecs), dtype
        assert np.array_equal(hh.cpu().numpy().view(np.uint32), eh), dtype
    assert torch.equal(ps, ops.class_prob_sum(zt, invT))
    ss, h2 = ops.bvsb_region_accum(zt, st, None, S, invT)
    assert torch.equal(h2, hh) and torch.equal(cs.sum(dim=2), ss)
    w = (np.random.RandomState(9).uniform(0.2, 1.0, size=C)).astype(np.float32)
    for weights in (np.ones(C, dtype=np.float32), w):
        w31 = ops.weights_to_fixed31(weights)
        score, dom, cnt, h64 = ops.region_finalize_weighted(cs, hh, torch.from_numpy(w31.view(np.int32)).cuda(), C - 1, True)
        escore, edom, ecnt = exact.region_finalize_weighted(ecs, eh, w31, C - 1)
        assert np.array_equal(score.cpu().numpy(), escore) and np.array_equal(dom.cpu().numpy(), edom)
        assert np.array_equal(cnt.cpu().numpy().view(np.uint32), ecnt)
    one = torch.from_numpy(ops.weights_to_fixed31(np.ones(C, dtype=np.float32)).view(np.int32)).cuda()
    s1 = ops.region_finalize_weighted(cs, hh, one, -1)[0]
    s2 = ops.region_finalize(ss, h2, -1)[0]
    assert torch.equal(s1, s2)


def test_single_pass_full_resolution_properties():
    """Size-independent checks at the bench shape [4,20,1024,2048], S=2048: every pixel lands in exactly one
    (region, class) bin, class-probability quanta sum to one per pixel, and a second accumulation doubles everything
    (integer accumulators: exact linearity)."""
    ops = _gpu()
    B, C, H, W, S = 4, 20, 1024, 2048, 2048
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    z = 0.4 * torch.randn((B, C, H, W), generator=g, device='cuda')
    spx = torch.from_numpy(np.stack([synth.superpixel_map(70 + i, H, W, S) for i in range(B)])).cuda()
    invT = ops.inv_temperature(0.1)
    ps, cs, hh = ops.single_pass_accum(z, spx, S, invT)
    assert int(hh.sum()) == B * H * W
    tot = ps.sum(dim=1).double() / 2.0 ** 23 / (H * W)
    assert float((tot - 1).abs().max()) < 1e-6
    ps2, cs2, hh2 = ops.single_pass_accum(z, spx, S, invT, prob_sum=ps.clone(), class_sum=cs.clone(), hist=hh.clone())
    assert torch.equal(ps2, 2 * ps) and torch.equal(cs2, 2 * cs) and torch.equal(hh2, 2 * hh)
    # permuting the images permutes the outputs (no cross-image leakage)
    perm = torch.tensor([2, 0, 3, 1], device='cuda')
    ps3, cs3, hh3 = ops.single_pass_accum(z[perm].contiguous(), spx[perm].contiguous(), S, invT)
    assert torch.equal(ps3, ps[perm]) and torch.equal(cs3, cs[perm]) and torch.equal(hh3, hh[perm])


def test_table_overflow_falls_back_to_global_atomics():
    """Adversarial id maps: every pixel of a tile carries a different superpixel id (far more ids than LDS table slots),
    plus a map with a single id -- both paths (LDS table, direct global atomics) must give the oracle's bits."""
    ops = _gpu()
    from oracle import exact
    _table_overflow_case(ops, exact, 32)
    _table_overflow_case(ops, exact, 64)        # tall enough for the ring kernel


def _table_overflow_case(ops, exact, H):
    B, C, W = 1, 20, 512
    S = H * W
    z = synth.logits(9, B, C, H, W)
    invT = ops.inv_temperature(0.1)
    rs = np.random.RandomState(1)
    for spx in (rs.permutation(S).reshape(B, H, W).astype(np.int64), np.zeros((B, H, W), dtype=np.int64),
                rs.randint(0, 300, size=(B, H, W)).astype(np.int64)):
        zt, st = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda()
        es, eh = exact.bvsb_region_accum(z, spx, None, S, np.float32(invT))
        gs, gh = ops.bvsb_region_accum(zt, st, None, S, invT)
        assert np.array_equal(gs.cpu().numpy().view(np.uint64), es) and np.array_equal(gh.cpu().numpy().view(np.uint32), eh)
        eps_, ecs, eh2 = exact.single_pass_accum(z, spx, S, np.float32(invT))
        ps, cs, hh = ops.single_pass_accum(zt, st, S, invT)
        assert np.array_equal(ps.cpu().numpy().view(np.uint64), eps_)
        assert np.array_equal(cs.cpu().numpy().view(np.uint64), ecs) and np.array_equal(hh.cpu().numpy().view(np.uint32), eh2)


def test_extreme_logits_stay_finite_and_exact():
    """Large-magnitude logits (|z|/T up to 500): the clamped exp keeps every probability finite; bits still match."""
    ops = _gpu()
    from oracle import exact
    B, C, H, W, S = 1, 20, 16, 256, 8
    z, spx = _case(77, B, C, H, W, S)
    z = (z * 30.0).astype(np.float32)
    invT = ops.inv_temperature(0.1)
    zt, st = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda()
    eps_, ecs, eh = exact.single_pass_accum(z, spx, S, np.float32(invT))
    ps, cs, hh = ops.single_pass_accum(zt, st, S, invT)
    assert np.array_equal(ps.cpu().numpy().view(np.uint64), eps_) and np.array_equal(cs.cpu().numpy().view(np.uint64), ecs)
    tot = ps.cpu().numpy().view(np.uint64).sum(axis=1).astype(np.float64) / 2.0 ** 23 / (H * W)
    assert np.all(np.abs(tot - 1.0) < 1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_randomised_shapes_and_dirty_ids(seed):
    """Random shapes around the kernel-selection boundaries (ring kernel for W % 4 == 0 and H >= 64, one-row kernel
    otherwise; 19/20/21 or runtime channel counts), random noise ids, ids outside [0, S) and negative ids:
    all three scan kernels equal the C oracle bit for bit."""
    ops = _gpu()
    from oracle import exact
    rs = np.random.RandomState(1000 + seed)
    B = int(rs.randint(1, 3))
    C = int(rs.choice([19, 20, 21, 5, 32]))
    H = int(rs.choice([rs.randint(3, 64), rs.randint(64, 150)]))
    W = int(rs.choice([4 * rs.randint(1, 150), rs.randint(5, 300)]))
    S = int(rs.randint(2, 400))
    z = (rs.standard_normal((B, C, H, W)) * rs.choice([0.3, 1.0, 3.0])).astype(np.float32)
    spx = np.stack([synth.superpixel_map(seed * 7 + i, H, W, S) for i in range(B)]).astype(np.int64)
    dirty = rs.uniform(size=spx.shape)
    spx[dirty < 0.03] = -1                   # invalid: skipped
    spx[(dirty >= 0.03) & (dirty < 0.06)] = S + int(rs.randint(0, 5))
    spx[(dirty >= 0.06) & (dirty < 0.15)] = rs.randint(0, S, size=int(((dirty >= 0.06) & (dirty < 0.15)).sum()))
    invT = ops.inv_temperature(float(rs.choice([0.1, 1.0])))
    w = rs.uniform(0.2, 1.0, size=C).astype(np.float32)
    zt, st = torch.from_numpy(z).cuda(), torch.from_numpy(spx).cuda()
    eps_, ecs, eh = exact.single_pass_accum(z, spx, S, np.float32(invT))
    for dtype in (torch.int64, torch.int32):
        ps, cs, hh = ops.single_pass_accum(zt, st.to(dtype), S, invT)
        assert np.array_equal(ps.cpu().numpy().view(np.uint64), eps_), (B, C, H, W, S, dtype)
        assert np.array_equal(cs.cpu().numpy().view(np.uint64), ecs), (B, C, H, W, S, dtype)
        assert np.array_equal(hh.cpu().numpy().view(np.uint32), eh), (B, C, H, W, S, dtype)
    es, eh3 = exact.bvsb_region_accum(z, spx, w, S, np.float32(invT))
    gs, gh = ops.bvsb_region_accum(zt, st, torch.from_numpy(w).cuda(), S, invT)
    assert np.array_equal(gs.cpu().numpy().view(np.uint64), es) and np.array_equal(gh.cpu().numpy().view(np.uint32), eh3)
    assert np.array_equal(ops.class_prob_sum(zt, invT).cpu().numpy().view(np.uint64), eps_)


@pytest.mark.parametrize("n_img,batch,C,n_batches", [(7, 2, 20, 4), (2975, 4, 20, 744), (5000, 3, 21, 1667), (9, 4, 19, 5), (1, 1, 2, 1)])
def test_class_weight_kernel_bit_exact(n_img, batch, C, n_batches):
    """k_class_weight (device f64, batch means added in batch order) == oracle/exact.c:exact_class_weight bit for bit:
    ragged last batch, more than one LDS chunk of batches (1 667 > 1 024), a trailing batch without pictures (9 pictures
    in 5 batches of 4: the gather pad of engine.ShardPlan), and the integer form floor(w * 2^31)."""
    from mulactseg_amd import ops
    from oracle import exact
    rs = np.random.RandomState(n_img)
    hw = 1024 * 2048
    ps = (rs.rand(n_img, C) * hw * 8388608.0 / C * 2).astype(np.uint64)
    ps[0, 0] = 0
    cum, w, w31 = ops.class_weight(torch.from_numpy(ps.view(np.int64)).cuda(), hw, batch, n_batches, 6.0)
    ecum, ew = exact.class_weight(ps, hw, (np.arange(n_img) // batch).astype(np.int32), n_batches, 6.0)
    assert np.array_equal(cum.cpu().numpy().view(np.uint64), ecum.view(np.uint64))
    assert np.array_equal(w.cpu().numpy().view(np.uint32), ew.view(np.uint32))
    assert np.array_equal(w31.cpu().numpy().view(np.uint32), exact.weights_to_fixed31(ew))
    from mulactseg_amd.active_selection.engine import class_weight_from_sums
    hcum, hw_ = class_weight_from_sums(ps.view(np.int64), hw, np.arange(n_img) // batch, n_batches, 6.0)
    assert np.array_equal(hcum.view(np.uint64), ecum.view(np.uint64)) and np.array_equal(hw_, ew)


@pytest.mark.parametrize("B,C,h,w,H,W,S,dt", [(2, 20, 16, 64, 64, 256, 64, 'int64'), (1, 20, 35, 130, 140, 520, 300, 'int32'),
                                              (1, 19, 193, 193, 769, 769, 2048, 'int16'), (2, 21, 12, 9, 48, 36, 40, "int64"),
                                              (1, 20, 256, 512, 1024, 2048, 2048, 'int16')])
def test_lowres_scan_equals_the_scan_of_the_upsampled_logits(B, C, h, w, H, W, S, dt):
    """K8: mas_single_pass_accum_lowres(zq) == mas_single_pass_accum(upsample_bilinear(zq)) bit for bit (class-probability sums,
    per-(region, class) margin sums, histograms) -- odd sizes, the 769 / 193 ratio, partial tiles, all id types; and against
    oracle/exact.c on the C-upsampled tensor for the small cases."""
    from mulactseg_amd import ops
    zq = synth.logits(11 + h, B, C, h, w)
    spx = np.stack([synth.superpixel_map(300 + i, H, W, S) for i in range(B)])
    invT = ops.inv_temperature(0.1)
    zt = torch.from_numpy(zq).cuda()
    st = torch.from_numpy(spx).cuda().to(getattr(torch, dt))
    full = ops.upsample_bilinear(zt, (H, W))
    a = ops.single_pass_accum(full, st, S, invT)
    b = ops.single_pass_accum_lowres(zt, (H, W), st, S, invT)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # at the exact x4 ratio the kernel reads one period of the tap pattern per lane (8 LDS reads per class instead of 16): the generic
    # tap reads must give the same bits
    g = ops.single_pass_accum_lowres(zt, (H, W), st, S, invT, generic=True)
    for x, y in zip(g, b):
        assert torch.equal(x, y)
    if H * W <= 520 * 140:
        from oracle import exact
        e = exact.single_pass_accum(exact.upsample_bilinear(zq, H, W), spx, S, np.float32(invT))
        assert np.array_equal(b[0].cpu().numpy().view(np.uint64), e[0])
        assert np.array_equal(b[1].cpu().numpy().view(np.uint64), e[1])
        assert np.array_equal(b[2].cpu().numpy().view(np.uint32), e[2])


def test_lowres_scan_beside_the_convolutions_of_another_stream_gives_the_same_bits():
    """Round 6: the x4 form of the scan shared compute units with k_conv_bx of a second stream and miscounted (lanes 48-63 of a packed-f32
    multiply whose VGPR src1 was read with op_sel[1] = 1, next to another kernel's MFMA waves: NOTEBOOK.md section 16.7).  The kernel now
    selects on src0 (common.h: mas_pk_mul_lo / _hi); quad-sized regions make every pixel's arg-max class and margin visible."""
    from mulactseg_amd import ops
    B, C, H, W = 2, 20, 256, 512
    S = H * W // 4
    zt = torch.from_numpy(synth.logits(77, B, C, H // 4, W // 4)).cuda()
    st = (torch.arange(H * W, device='cuda', dtype=torch.int32) // 4).view(1, H, W).expand(B, H, W).contiguous()
    invT = ops.inv_temperature(0.1)
    conv = torch.nn.Conv2d(512, 512, 3, padding=2, dilation=2, bias=False).cuda()
    x = torch.randn((4, 512, 32, 64), device='cuda')

    def neighbour():
        with torch.no_grad():
            for _ in range(30):
                ops.conv_bx(conv, x)
    ref = ops.single_pass_accum_lowres(zt, (H, W), st, S, invT)
    neighbour()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for generic in (False, True):
        for rep in range(6):
            acc = (torch.zeros_like(ref[0]), torch.zeros_like(ref[1]), torch.zeros_like(ref[2]))
            torch.cuda.synchronize()
            with torch.cuda.stream(sb):
                neighbour()
            with torch.cuda.stream(sa):
                ops.single_pass_accum_lowres(zt, (H, W), st, S, invT, prob_sum=acc[0], class_sum=acc[1], hist=acc[2], generic=generic)
            with torch.cuda.stream(sb):
                neighbour()
            torch.cuda.synchronize()
            for got, want in zip(acc, ref):
                assert torch.equal(got, want), (generic, rep)


def test_lowres_scan_refuses_ratios_it_has_no_footprint_for():
    from mulactseg_amd import _lib, ops
    zq = torch.zeros((1, 20, 32, 32), device='cuda')
    spx = torch.zeros((1, 64, 64), dtype=torch.int64, device='cuda')
    with pytest.raises(_lib.MulActSegHipError):
        ops.single_pass_accum_lowres(zq, (64, 64), spx, 8, 10.0)
