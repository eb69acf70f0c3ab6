"""csrc/conv_bx.hip: the inference convolutions on the bf16 matrix cores with f32 operands and results (every operand split
exactly into three bf16 terms, six partial products, f32 accumulation) against torch's conv2d in float64 on the layer
geometries of the network (models/segmentation/backbone/resnet.py:129-160, deeplabv3.py:85-137): 1x1 at stride 1 / 2, 3x3 at
dilation 1 / 2, partial tiles, planes that are not a multiple of the tile, every epilogue.  The error bound is the one of
the f32-MFMA kernel (tests/test_conv_mfma_gpu.py): 2e-5 of the output scale -- and the two kernels are compared with each
other on the same data: the split form must not be less accurate than the f32 form by more than rounding noise."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Cin, Cout, k, stride, dil, N, H, W
    (128, 256, 1, 1, 1, 2, 24, 40),      # layer1 downsample shape: 128-row M tile, partial pixel tile
    (256, 64, 1, 1, 1, 1, 20, 52),       # layer1 conv1: 64-row M tile x 256 pixels
    (64, 256, 1, 1, 1, 2, 16, 48),       # layer1 conv3: two chunks
    (1024, 512, 1, 1, 1, 1, 12, 12),     # deep 1x1: 32 chunks, one partial tile
    (256, 512, 1, 2, 1, 2, 34, 64),      # downsample: 1x1 stride 2
    (512, 1024, 1, 2, 1, 1, 16, 40),     # downsample, plane narrower than a tile row
    (2048, 256, 1, 1, 1, 1, 9, 36),      # ASPP 1x1
    (32, 200, 1, 1, 1, 1, 10, 30),       # Cout 200: the last 64-row M tile is mostly padding
    (64, 64, 3, 1, 1, 2, 20, 70),        # layer1 conv2
    (64, 128, 3, 1, 1, 1, 33, 45),       # stem conv3, odd plane
    (512, 512, 3, 1, 2, 1, 13, 40),      # layer4 conv2: dilation 2
    (256, 256, 3, 1, 2, 1, 9, 33),       # dilation 2, partial tiles in both directions
    (8, 64, 3, 1, 1, 1, 5, 32),          # one chunk, plane lower than a tile
    (64, 200, 3, 1, 1, 1, 9, 37),        # Cout 200
    (304, 256, 1, 1, 1, 1, 16, 48),      # decoder pointwise: Cin = 9.5 chunks of 32 (the last one zero-padded)
    (50, 64, 1, 1, 1, 2, 8, 32),         # Cin not a multiple of anything
    (20, 64, 3, 1, 1, 1, 11, 40),        # 3x3 with 2.5 chunks of 8 channels
    (64, 256, 1, 1, 1, 2, 25, 33),       # odd plane (the 769-crop planes 385 / 193 / 97 / 49 scaled down): quads run over plane ends
    (256, 128, 1, 1, 1, 1, 49, 49),
    (64, 64, 3, 1, 1, 1, 49, 49),        # 3x3 on a 49-wide plane: the second tile column is half padding
    (64, 1024, 1, 1, 1, 4, 64, 128),     # many tiles: 2 048 workgroups, four rounds
    (48, 1024, 1, 1, 1, 4, 60, 130),     # Cin = 1.5 chunks, partial pixel tile, many M tiles
    (128, 128, 3, 2, 1, 2, 24, 40),      # layer2.0.conv2: 3x3 stride 2 as nine shifted 1x1 stride-2 products (36 chunks), partial pixel tile
    (256, 256, 3, 2, 1, 1, 18, 64),      # layer3.0.conv2
    (64, 200, 3, 2, 1, 1, 10, 16),       # 64-row M tile (256 output pixels per tile), Cout 200, every output row starts a quad
    (32, 64, 3, 2, 1, 3, 6, 8),          # one chunk per tap, 3 x 4 outputs: every quad touches the top or the left edge
    (128, 128, 3, 2, 1, 4, 64, 128),     # many tiles
]


def _ref(x, conv, bn, relu, res):
    y = conv(x)
    if bn is not None:
        y = bn(y)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", CASES)
@pytest.mark.parametrize("epi", ["bare", "bn_relu", "bn_res_relu"])
def test_conv_bx_matches_conv2d(Cin, Cout, k, stride, dil, N, H, W, epi):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin * 7 + Cout + k + stride + dil + H)
    conv = nn.Conv2d(Cin, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = None
    if epi != "bare":
        bn = nn.BatchNorm2d(Cout).cuda().eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(N, Cin, H, W, device='cuda')
    assert ops.conv_bx_supported(conv, x)
    with torch.no_grad():
        res = torch.randn_like(conv(x)) if epi == "bn_res_relu" else None
        ref = _ref(x.double(), conv.double(), bn.double() if bn is not None else None, epi != "bare", res.double() if res is not None else None)
        conv.float()
        if bn is not None:
            bn.float()
        y = ops.conv_bx(conv, x, bn, relu=epi != "bare", residual=res)
        y32 = ops.conv_mfma(conv, x, bn, relu=epi != "bare", residual=res) if ops.conv_mfma_supported(conv, x) else None
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    if y32 is not None:                                                 # as accurate as the f32 matrix cores on the same data
        err32 = float((y32.double() - ref).abs().max())
        assert err <= 2.0 * err32 + 1e-6 * scale, (err, err32, scale)


def test_conv_bx_exact_on_integers():
    """Integers below 2^8 are single bf16 terms and their products and sums are exact in f32: lane maps, tap offsets, the
    zero tap, the position swizzle of the 1x1 image, the chunk order and the epilogue are pinned bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    for k, stride, dil, cout in ((1, 1, 1, 128), (1, 1, 1, 64), (3, 1, 1, 128), (3, 1, 2, 64), (1, 2, 1, 128), (1, 2, 1, 64), (1, 1, 1, 1024),
                                 (3, 2, 1, 128), (3, 2, 1, 64)):
        conv = nn.Conv2d(64, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
        with torch.no_grad():
            conv.weight.copy_(torch.randint(-3, 4, conv.weight.shape, generator=g, device='cuda').float())
            shape = (4, 64, 64, 128) if cout == 1024 else (2, 64, 38, 56)          # (1024 channels on 256 pixel tiles: four rounds of workgroups)
            x = torch.randint(-4, 5, shape, generator=g, device='cuda').float()
            assert ops.conv_bx_supported(conv, x)
            y = ops.conv_bx(conv, x)
            ref = F.conv2d(x.double(), conv.weight.double(), None, stride, dil if k == 3 else 0, dil).float()
            assert torch.equal(y, ref), (k, stride, dil, cout)


def test_conv_bx_keeps_all_24_bits_of_both_operands():
    """Operands with full 24-bit significands whose products need the low terms: integers up to 2^24 times powers of two chosen so
    that every partial sum is an integer below 2^24 ONLY if the (h, m, l) terms are all there -- x = 2^16 a + 2^8 b + c against
    weights of 1: the sum over 32 channels of x must come back exactly (a one-term or two-term split loses c or b)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    conv = nn.Conv2d(32, 64, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.zero_()
        conv.weight[:, 0] = 1.0                                   # y[m] = x[channel 0]: the product 1 * x must keep all of x
        conv.weight[1, :, 0, 0] = 0.0
        conv.weight[1, 5] = 3.0                                   # 3 * x[5]: h*h, m*h, l*h terms with a two-bit weight
        a = torch.randint(1 << 23, 1 << 24, (1, 32, 8, 32), generator=g, device='cuda')
        x = a.float()                                             # full 24-bit significands
        y = ops.conv_bx(conv, x)
        assert torch.equal(y[:, 0], x[:, 0])
        assert torch.equal(y[:, 1].double(), (3.0 * x[:, 5].double()).float().double()) or \
            float((y[:, 1].double() - 3.0 * x[:, 5].double()).abs().max()) <= 4.0        # 3x needs 26 bits: one rounding (ulp = 4 at 2^25)
        # and the weight side: w with 24 bits, x = 1
        w = torch.randint(1 << 23, 1 << 24, (64,), generator=g, device='cuda').float()
        conv.weight.zero_()
        conv.weight[:, 7, 0, 0] = w
        x.zero_()
        x[:, 7] = 1.0
        y = ops.conv_bx(conv, x)
        assert torch.equal(y[0, :, 3, 3], w)


def test_conv_bx_weight_cache_follows_the_parameter():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    conv = nn.Conv2d(64, 64, 1, bias=False).cuda()
    x = torch.randn(1, 64, 8, 32, device='cuda')
    with torch.no_grad():
        a = ops.conv_bx(conv, x)
        conv.weight.mul_(2.0)
        b = ops.conv_bx(conv, x)
    assert torch.allclose(b, 2 * a, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("seed", range(28))
def test_conv_bx_randomised_geometries(seed):
    """Random supported geometries against conv2d in float64: partial tiles, halo clamps, the chunk walk, the padding of the
    output channels hold wherever conv_bx_supported says yes."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    rs = np.random.RandomState(2000 + seed)
    k = int(rs.choice([1, 3]))
    stride = int(rs.choice([1, 2])) if (k == 1 or seed >= 16) else 1          # (seeds 16..: the strided 3x3 too)
    dil = 1 if (k == 1 or stride == 2) else int(rs.choice([1, 2]))
    cin = int(rs.choice([8, 12, 16, 24, 40, 64, 72, 128])) if k == 3 else int(rs.choice([32, 48, 64, 96, 100, 160]))
    if k == 3 and stride == 2:
        cin = int(rs.choice([32, 64, 96, 128]))
    cout = int(rs.choice([16, 48, 64, 80, 128, 192, 200, 256]))
    N = int(rs.randint(1, 4))
    if k == 3 and stride == 1:
        H, W = int(rs.randint(3, 40)), int(rs.randint(32, 90))
    elif stride == 2:
        H, W = 2 * int(rs.randint(2, 30)), 8 * int(rs.randint(1, 12))
    else:
        H, W = int(rs.randint(3, 60)), 4 * int(rs.randint(1, 24))
    torch.manual_seed(seed)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, cin, H, W, device='cuda')
        assert ops.conv_bx_supported(conv, x), (k, stride, dil, cin, cout, H, W)
        res = torch.randn_like(conv(x)) if seed % 2 else None
        ref = _ref(x.double(), conv.double(), bn.double(), True, res.double() if res is not None else None)
        conv.float(); bn.float()
        y = ops.conv_bx(conv, x, bn, relu=True, residual=res)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (k, stride, dil, cin, cout, N, H, W)


WGRAD_CASES = [
    # N, Cin, Cout, H, W
    (2, 128, 256, 16, 32),       # one tile column, two tile rows
    (1, 256, 128, 8, 48),
    (3, 304, 256, 8, 16),        # Cin 304: the third column tile is mostly padding
    (2, 512, 2048, 12, 16),      # many tiles
    (4, 1280, 256, 4, 8),        # one chunk per picture
    (2, 100, 72, 8, 8),          # both extents below one tile
    (2, 128, 256, 25, 33),       # odd plane: every picture ends in a partial, masked chunk; rows start at any 4-byte alignment
    (1, 256, 128, 49, 49),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", WGRAD_CASES)
def test_wgrad_bx_matches_float64(N, Cin, Cout, H, W):
    """csrc/conv_wgrad_bx.hip against the float64 product, and against the f32 matrix-core kernel (csrc/conv_wgrad.hip) on the same
    data: the split form is as accurate; two runs give identical bits (fixed-order split-K reduction)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(N * 1000 + Cin + Cout)
    x = torch.randn(N, Cin, H, W, device='cuda')
    dy = torch.randn(N, Cout, H, W, device='cuda')
    ref = torch.einsum('nmp,ncp->mc', dy.double().flatten(2), x.double().flatten(2))
    dw = ops.conv_wgrad_bx(x, dy)
    assert dw.shape == (Cout, Cin, 1, 1)
    dw32 = ops.conv_wgrad(x, dy, 1, 1, 1) if Cout < 96 else None      # (below 96 output channels conv_wgrad keeps the f32 kernel)
    scale = float(ref.abs().max())
    err = float((dw[:, :, 0, 0].double() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    if dw32 is not None:
        err32 = float((dw32[:, :, 0, 0].double() - ref).abs().max())
        assert err <= 2.0 * err32 + 1e-6 * scale, (err, err32)
    assert torch.equal(dw, ops.conv_wgrad_bx(x, dy))


def test_wgrad_bx_exact_on_integers_and_selected_by_conv_wgrad():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randint(-3, 4, (2, 256, 16, 24), generator=g, device='cuda').float()
    dy = torch.randint(-2, 3, (2, 128, 16, 24), generator=g, device='cuda').float()
    ref = torch.einsum('nmp,ncp->mc', dy.double().flatten(2), x.double().flatten(2)).float()
    assert torch.equal(ops.conv_wgrad_bx(x, dy)[:, :, 0, 0], ref)
    assert torch.equal(ops.conv_wgrad(x, dy, 1, 1, 1)[:, :, 0, 0], ref)


@pytest.mark.parametrize("shape,role", [((130, 40, 1, 1), 0), ((130, 40, 1, 1), 1), ((64, 20, 3, 3), 0), ((72, 16, 3, 3), 1), ((256, 304, 1, 1), 0),
                                        ((128, 64, 3, 3), 2), ((200, 32, 3, 3), 2)])       # role 2: tap-major chunks of the strided 3x3
def test_weight_image_equals_the_numpy_restatement(shape, role):
    """mas_conv_bx_pack (and the multi-job pack the training step uses) against oracle/bx_split.py:pack_image, bit for bit: the
    split of every weight, the zero padding of rows / channels / the tenth tap, the mirrored taps of the input-gradient role."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    from oracle import bx_split
    rs = np.random.RandomState(sum(shape) + role)
    w = rs.standard_normal(shape).astype(np.float32)
    wt = torch.from_numpy(w).cuda()
    want = bx_split.pack_image(w, role)
    assert np.array_equal(ops.conv_bx_pack(wt, role).cpu().numpy().view(np.uint16), want)
    img = ops.bx_packed_weight(wt, role)                    # registry: packs alone the first time ...
    assert np.array_equal(img.cpu().numpy().view(np.uint16), want)
    wt.mul_(0.5)                                            # ... and through ONE multi-job launch when a version counter moved
    img = ops.bx_packed_weight(wt, role)
    assert np.array_equal(img.cpu().numpy().view(np.uint16), bx_split.pack_image(w * np.float32(0.5), role))


@pytest.mark.parametrize("Ca,Cb,Cout,N,H,W", [(64, 128, 256, 2, 24, 40), (512, 1024, 2048, 1, 12, 20), (48, 100, 192, 2, 9, 37)])
def test_conv_bx_dual_matches_the_two_convolutions(Ca, Cb, Cout, N, H, W):
    """mas_conv_bx_fwd_dual: relu(bn_a(conv_a(xa)) + bn_b(conv_b(xb))) -- conv3 + stride-1 downsample of a Bottleneck in one kernel,
    both BatchNorm scales folded into the weight images -- against float64, and against the two-kernel form (downsample, then conv3
    with the residual operand); the folded weight image against the numpy restatement bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    from oracle import bx_split
    torch.manual_seed(Ca + Cb + Cout)
    conv_a, conv_b = nn.Conv2d(Ca, Cout, 1, bias=False).cuda(), nn.Conv2d(Cb, Cout, 1, bias=False).cuda()
    bn_a, bn_b = nn.BatchNorm2d(Cout).cuda().eval(), nn.BatchNorm2d(Cout).cuda().eval()
    with torch.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        xa, xb = torch.randn(N, Ca, H, W, device='cuda'), torch.randn(N, Cb, H, W, device='cuda')
        assert ops.conv_bx_dual_supported(conv_a, xa, conv_b, xb)
        ref = F.relu(bn_a.double()(conv_a.double()(xa.double())) + bn_b.double()(conv_b.double()(xb.double())))
        for m in (conv_a, conv_b, bn_a, bn_b):
            m.float()
        y = ops.conv_bx_dual(conv_a, bn_a, xa, conv_b, bn_b, xb, True)
        two = ops.conv_bx(conv_a, xa, bn_a, relu=True, residual=ops.conv_bx(conv_b, xb, bn_b, relu=False))
        scale_a, _ = ops._bn_fold(bn_a)
        img = ops.conv_bx_pack(conv_a.weight.detach(), 0, row_scale=scale_a).cpu().numpy().view(np.uint16)
    assert np.array_equal(img, bx_split.pack_image(conv_a.weight.detach().cpu().numpy(), 0, row_scale=scale_a.cpu().numpy()))
    sc = float(ref.abs().max())
    err, err2 = float((y.double() - ref).abs().max()), float((two.double() - ref).abs().max())
    assert err <= 2e-5 * sc, (err, sc)
    assert err <= 2.0 * err2 + 2e-6 * sc, (err, err2)


SPLIT_CASES = [
    # Cin, Cout, k, dil, N, H, W -- the layers of the 48 x 48 (49 x 49) planes of the training crop, scaled where a full one would only add time
    (1024, 256, 1, 1, 4, 48, 48),       # layer3 conv1: 144 workgroups for 512 slots
    (2048, 512, 1, 1, 2, 48, 48),
    (1280, 256, 1, 1, 2, 49, 49),       # ASPP projection on the 769-crop plane (odd plane: out % 4 != 0 falls back to one part)
    (256, 256, 3, 1, 4, 48, 48),        # layer3 conv2: 16 x 16 tiles, no padded quarter
    (512, 512, 3, 2, 2, 48, 48),        # layer4 conv2, dilation 2
    (256, 256, 3, 1, 1, 49, 49),
    (128, 128, 3, 1, 2, 97, 97),        # 97 x 97: 49 square tiles against 52 wide ones
    (72, 64, 3, 1, 1, 20, 36),          # 9 chunks of 8 channels, partial tiles of both shapes
    (256, 256, 3, 2, 1, 49, 49),        # dilation 2 on the 49 x 49 plane: flat tiles with an 11-row patch
    (64, 64, 3, 1, 2, 23, 37),          # flat tiles whose runs start and end in the middle of rows
    (200, 136, 1, 1, 1, 24, 40),        # Cin = 6.25 chunks, Cout = one M tile + 8 rows
]


@pytest.mark.parametrize("Cin,Cout,k,dil,N,H,W", SPLIT_CASES)
def test_split_k_and_square_tiles_match_conv2d(Cin, Cout, k, dil, N, H, W):
    """mas_conv_bx_train: every (ksplit, tile shape) of a product gives the float64 result at the bar of the unsplit kernel, both
    roles (forward; input gradient + the other consumer's gradient as residual), run-to-run identical; the library's own plan is
    one of them."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin + 3 * Cout + k + dil + H)
    w = torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5
    x = torch.randn(N, Cin, H, W, device='cuda')
    dy = torch.randn(N, Cout, H, W, device='cuda')
    other = torch.randn(N, Cin, H, W, device='cuda')
    pad = dil if k == 3 else 0
    ref_y = F.conv2d(x.double(), w.double(), None, 1, pad, dil)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, pad, dil) + other.double()
    plan = ops.conv_bx_train_plan(x.shape, w.shape, dil, False)
    assert plan[0] >= 1
    pk0, pk1 = ops.conv_bx_pack(w, 0), ops.conv_bx_pack(w, 1)
    nch = -(-Cin // (32 if k == 1 else 8))
    flat_ok = k == 3 and W >= 8 and (((256 + W - 2) // W + 1) + 2 * dil) * (W + 2 * dil) <= 608
    assert plan[1] in (16, 32) or (plan[1] == 1 and flat_ok)
    for ks in sorted({1, 2, 3, min(5, nch), plan[0]}):
        if ks > nch:
            continue
        for tw in (((32, 16, 1) if flat_ok else (32, 16)) if k == 3 else (32,)):
            y = ops.conv_bx_raw(x, w, dil, packed=pk0, ksplit=ks, tile_w=tw)
            err = float((y.double() - ref_y).abs().max())
            assert err <= 2e-5 * float(ref_y.abs().max()), (ks, tw, err)
            assert torch.equal(y, ops.conv_bx_raw(x, w, dil, packed=pk0, ksplit=ks, tile_w=tw)), (ks, tw)
    ncd = -(-Cout // (32 if k == 1 else 8))
    for ks in sorted({1, 2, min(4, ncd)}):
        dx = ops.conv_bx_raw(dy, w, dil, dgrad=True, residual=other, packed=pk1, ksplit=ks, tile_w=(1 if flat_ok else 16) if k == 3 else 32)
        err = float((dx.double() - ref_dx).abs().max())
        assert err <= 2e-5 * float(ref_dx.abs().max()), (ks, err)
    y_auto = ops.conv_bx_raw(x, w, dil, packed=pk0)
    assert float((y_auto.double() - ref_y).abs().max()) <= 2e-5 * float(ref_y.abs().max())


def test_split_k_exact_on_integers():
    """Integer data: every partial tile and their sum are exact, so all plans give the same bits as float64 conv2d."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(11)
    for (Cin, Cout, k, dil, N, H, W) in ((512, 128, 1, 1, 2, 48, 48), (128, 64, 3, 1, 2, 48, 48), (96, 128, 3, 2, 1, 33, 50)):
        w = torch.randint(-3, 4, (Cout, Cin, k, k), generator=g, device='cuda').float()
        x = torch.randint(-4, 5, (N, Cin, H, W), generator=g, device='cuda').float()
        res = torch.randint(-9, 10, (N, Cout, H, W), generator=g, device='cuda').float()
        ref = F.conv2d(x.double(), w.double(), None, 1, dil if k == 3 else 0, dil).float()
        flat_ok = k == 3 and (((256 + W - 2) // W + 1) + 2 * dil) * (W + 2 * dil) <= 608
        for ks in (1, 2, 3, 4):
            for tw in (((32, 16, 1) if flat_ok else (32, 16)) if k == 3 else (32,)):
                assert torch.equal(ops.conv_bx_raw(x, w, dil, ksplit=ks, tile_w=tw), ref), (Cin, k, ks, tw)
        # the residual operand rides on part 0 only
        wt = torch.randint(-3, 4, (Cin, Cout, k, k), generator=g, device='cuda').float()      # (dgrad role: x is the "dY" of a Cin <- Cout layer)
        ref_dx = torch.nn.grad.conv2d_input((N, Cout, H, W), wt.double(), x.double(), 1, dil if k == 3 else 0, dil).float() + res
        for ks in (1, 3):
            assert torch.equal(ops.conv_bx_raw(x, wt, dil, dgrad=True, residual=res, ksplit=ks), ref_dx), (Cin, k, ks)


def test_non_finite_and_tiny_operands_behave_as_documented():
    """ADVICE r4 (csrc/bx_split.h): NaN stays NaN; an Inf operand gives NaN (h = Inf, m = Inf - Inf) where the f32 pipe would propagate
    Inf -- both poison exactly the outputs that read the bad pixel and nothing else; operands below 2^-100 lose their third term
    only (relative error <= 2^-15 instead of 2^-23); zeros times garbage-free padding stay zeros."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(2)
    w = torch.randn(64, 32, 1, 1, device='cuda')
    x = torch.randn(1, 32, 8, 32, device='cuda')
    clean = ops.conv_bx_raw(x, w)
    for bad in (float('nan'), float('inf'), float('-inf')):
        xb = x.clone()
        xb[0, 3, 2, 5] = bad
        y = ops.conv_bx_raw(xb, w)
        assert bool(torch.isnan(y[0, :, 2, 5]).all())
        keep = torch.ones_like(y, dtype=torch.bool)
        keep[0, :, 2, 5] = False
        assert torch.equal(y[keep], clean[keep])
    tiny = x * 1e-35                                              # 2^-116: below the range the three-term split covers
    yt = ops.conv_bx_raw(tiny, w)
    ref = F.conv2d(tiny.double(), w.double())
    assert bool(torch.isfinite(yt).all())
    assert float((yt.double() - ref).abs().max()) <= 2.0 ** -15 * float(ref.abs().max())


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", [c for c in CASES if c[3] == 1])
def test_presplit_input_gives_the_same_bits(Cin, Cout, k, stride, dil, N, H, W):
    """mas_conv_bx_fwd_pre on the bx3 form of x (mas_bx3_split) == mas_conv_bx_fwd on x, bit for bit, with every epilogue: the split
    of an element does not depend on who computes it, and the products are accumulated in the same order.  The bx3 tensor itself is
    checked against the numpy restatement of the split (oracle/bx_split.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    from oracle import bx_split
    torch.manual_seed(Cin + 5 * Cout + k + H)
    conv = nn.Conv2d(Cin, Cout, k, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = nn.BatchNorm2d(Cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, Cin, H, W, device='cuda')
        x3 = ops.bx3_split(x)
        if Cin * H * W <= 1 << 20:
            hml = [bx_split.bf16_bits(t) for t in bx_split.split3(x.cpu().numpy())]        # three uint16 arrays shaped like x
            G = (Cin + 7) // 8
            want = np.zeros((N, G, 3, H * W, 8), dtype=np.uint16)
            for t in range(3):
                pad = np.zeros((N, G * 8, H * W), dtype=np.uint16)
                pad[:, :Cin] = hml[t].reshape(N, Cin, H * W)
                want[:, :, t] = pad.reshape(N, G, 8, H * W).transpose(0, 1, 3, 2)
            assert np.array_equal(x3.data.cpu().numpy().view(np.uint16), want)
        res = torch.randn(N, Cout, H, W, device='cuda')
        for kw in (dict(), dict(bn=bn, relu=True), dict(bn=bn, relu=True, residual=res)):
            assert torch.equal(ops.conv_bx_pre(conv, x3, **kw), ops.conv_bx(conv, x, **kw)), kw.keys()


WGRAD3_CASES = [
    # Cin, Cout, dil, N, H, W
    (64, 64, 1, 2, 48, 48),          # layer1 conv2 scaled: 1 x 2 tiles, whole chunks
    (128, 128, 1, 1, 24, 40),        # partial chunk columns (40 = 2.5 x 16)
    (256, 256, 1, 2, 12, 12),        # plane smaller than a chunk row, many tiles
    (512, 512, 2, 1, 13, 21),        # dilation 2, odd plane
    (64, 128, 1, 1, 49, 49),         # the 769-crop planes: rows that are only 4-byte aligned
    (64, 64, 1, 1, 97, 97),
    (40, 72, 1, 2, 18, 35),          # Cin = 1.25 tiles of 32 (the last quad partial), Cout = 1.125 tiles of 64
    (35, 64, 2, 1, 9, 30),           # Cin not a multiple of 4: the last channel quad has three channels
    (8, 24, 1, 1, 5, 7),             # everything smaller than one tile / chunk
    (64, 64, 1, 4, 96, 96),          # split K over many ranges
]


@pytest.mark.parametrize("Cin,Cout,dil,N,H,W", WGRAD3_CASES)
def test_wgrad_3x3_on_the_bf16_cores_matches_float64(Cin, Cout, dil, N, H, W):
    """mas_conv_wgrad_bx3 (X patch through ds_read_b64_tr_b16) against the float64 weight gradient of conv2d, at the bar of the f32
    kernel it replaces (2e-5 of the result's scale) and no worse than 2x that kernel's own error; run-to-run identical."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin + 7 * Cout + dil + H + W)
    x = torch.randn(N, Cin, H, W, device='cuda')
    dy = torch.randn(N, Cout, H, W, device='cuda')
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), 1, dil, dil)
    got = ops.conv_wgrad_bx3(x, dy, dil)
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    import os
    os.environ["MAS_WGRAD3"] = "f32"
    try:
        f32 = ops.conv_wgrad(x, dy, 3, 1, dil)
    finally:
        os.environ.pop("MAS_WGRAD3")
    err32 = float((f32.double() - ref).abs().max())
    assert err <= 2.0 * err32 + 1e-6 * scale, (err, err32, scale)
    assert torch.equal(got, ops.conv_wgrad_bx3(x, dy, dil))
    assert torch.equal(ops.conv_wgrad(x, dy, 3, 1, dil), got)             # the dispatcher takes this kernel


def test_wgrad_3x3_exact_on_integers_and_tap_by_tap():
    """Integer data: every product and partial sum is exact, so the result equals float64 bit for bit -- the lane maps of the
    transposing read, the tap offsets, the halo, the zero padding at the plane's edges and the chunk edges are all pinned.  And a
    one-hot probe: dY = a single 1 at (m, y, x) makes dW[m, c, ty, tx] = X[c, y + (ty-1) d, x + (tx-1) d] for every tap."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(31)
    for (Cin, Cout, dil, N, H, W) in ((64, 64, 1, 2, 20, 36), (32, 96, 2, 1, 11, 19), (96, 64, 1, 1, 8, 16), (64, 64, 2, 1, 49, 49)):
        x = torch.randint(-4, 5, (N, Cin, H, W), generator=g, device='cuda').float()
        dy = torch.randint(-3, 4, (N, Cout, H, W), generator=g, device='cuda').float()
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), 1, dil, dil).float()
        assert torch.equal(ops.conv_wgrad_bx3(x, dy, dil), ref), (Cin, Cout, dil, H, W)
    for dil in (1, 2):
        H, W = 10, 23
        x = torch.arange(1 * 32 * H * W, device='cuda', dtype=torch.float32).reshape(1, 32, H, W) % 251
        for (m, y, xx) in ((5, 0, 0), (40, 4, 15), (63, H - 1, W - 1), (17, 3, 16)):
            dy = torch.zeros(1, 64, H, W, device='cuda')
            dy[0, m, y, xx] = 1.0
            dw = ops.conv_wgrad_bx3(x, dy, dil)
            assert float(dw[torch.arange(64, device='cuda') != m].abs().max()) == 0.0
            for ty in range(3):
                for tx in range(3):
                    iy, ix = y + (ty - 1) * dil, xx + (tx - 1) * dil
                    want = x[0, :, iy, ix] if (0 <= iy < H and 0 <= ix < W) else torch.zeros(32, device='cuda')
                    assert torch.equal(dw[m, :, ty, tx], want), (dil, m, y, xx, ty, tx)


STAT_CASES = [
    # Cin, Cout, k, dil, N, H, W, tile_w
    (64, 256, 1, 1, 2, 48, 48, 32),      # 128-row M tiles, whole pixel tiles
    (256, 64, 1, 1, 2, 20, 36, 32),      # 64-row M tile x 256 pixels, partial pixel tile
    (32, 200, 1, 1, 1, 25, 33, 32),      # Cout 200: the last M tile is mostly padding; odd plane
    (64, 64, 3, 1, 2, 40, 72, 32),       # 3x3, 8 x 32 tiles, partial tiles in both directions
    (64, 128, 3, 1, 1, 48, 48, 16),      # 16 x 16 tiles
    (72, 64, 3, 2, 1, 49, 49, 1),        # flat tiles, dilation 2
    (128, 1024, 1, 1, 4, 24, 24, 32),    # many M tiles
]


@pytest.mark.parametrize("Cin,Cout,k,dil,N,H,W,tw", STAT_CASES)
def test_epilogue_batchnorm_partials_are_the_sums_of_the_stored_outputs(Cin, Cout, k, dil, N, H, W, tw):
    """mas_conv_bx_train(stats): [Cout, slots, 2] partial sums formed in the epilogue (v_permlane16_swap / DPP halving over the 32
    lanes of a row) -- summed over the slots they are sum y and sum y^2 of the y the same launch stored: exactly on integer data,
    to f32 rounding on real data; pixels beyond the plane and rows beyond Cout contribute nothing; y itself is unchanged."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(Cin + Cout + H)
    pad = dil if k == 3 else 0
    for integer in (True, False):
        if integer:
            w = torch.randint(-2, 3, (Cout, Cin, k, k), generator=g, device='cuda').float()
            x = torch.randint(-3, 4, (N, Cin, H, W), generator=g, device='cuda').float()
        else:
            w = torch.randn(Cout, Cin, k, k, generator=g, device='cuda') / (Cin * k * k) ** 0.5
            x = torch.randn(N, Cin, H, W, generator=g, device='cuda')
        y0 = ops.conv_bx_raw(x, w, dil, ksplit=1, tile_w=tw)
        y, part = ops.conv_bx_raw(x, w, dil, ksplit=1, tile_w=tw, stats=True)
        assert torch.equal(y, y0)
        assert part is not None and part.shape[0] == Cout and part.shape[2] == 2 and part.dtype == torch.float64
        s = part[:, :, 0].sum(1)
        q = part[:, :, 1].sum(1)
        ys = y.double().sum(dim=(0, 2, 3))
        yq = (y.double() ** 2).sum(dim=(0, 2, 3))
        if integer:
            assert torch.equal(s, ys) and torch.equal(q, yq), (float((s - ys).abs().max()), float((q - yq).abs().max()))
        else:
            assert float((s - ys).abs().max()) <= 1e-5 * float(y.abs().double().sum(dim=(0, 2, 3)).max())
            assert float((q - yq).abs().max()) <= 1e-5 * float(yq.max())
    # a plan that splits K forms them in its reduction pass (planes of a multiple of four pixels; None otherwise: the caller's
    # BatchNorm then runs its own reduction pass)
    nch = -(-Cin // (32 if k == 1 else 8))
    if nch >= 2:
        y2, part2 = ops.conv_bx_raw(x, w, dil, ksplit=2, tile_w=tw, stats=True)
        assert torch.equal(y2, ops.conv_bx_raw(x, w, dil, ksplit=2, tile_w=tw))
        if (H * W) % 4 == 0:
            assert part2 is not None and part2.shape == (Cout, N * (-(-H * W // 4096)), 2)
            ys2, yq2 = y2.double().sum(dim=(0, 2, 3)), (y2.double() ** 2).sum(dim=(0, 2, 3))
            assert float((part2[:, :, 0].sum(1) - ys2).abs().max()) <= 1e-5 * float(y2.abs().double().sum(dim=(0, 2, 3)).max())
            assert float((part2[:, :, 1].sum(1) - yq2).abs().max()) <= 1e-5 * float(yq2.max())
        else:
            assert part2 is None
