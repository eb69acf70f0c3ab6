"""Training-mode dense convolutions (csrc/conv_wgrad.hip + csrc/conv_mfma.hip through ops._ConvTrain): weight gradient,
input gradient and forward against float64 autograd of torch's conv2d on every layer geometry of the network
(models/segmentation/backbone/resnet.py:129-171, deeplabv3.py:85-137,168-245) -- 1x1 / 3x3, stride 1 / 2, dilation 1 / 2 / 4,
odd planes (the 769-crop sizes scaled down), channel counts that are not multiples of the tiles (3, 48, 304, 200), planes
smaller than one chunk.  v_mfma_f32_32x32x2_f32 is an exact-f32 fma chain, so the tolerance is that of a differently ordered
f32 sum: 2e-5 of the result's scale."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Cin, Cout, k, stride, dil, N, H, W
    (128, 64, 1, 1, 1, 2, 24, 40),       # layer1 conv1: 64-row M tile, flat chunks, vector loads
    (64, 256, 1, 1, 1, 1, 25, 33),       # odd plane -> element loads, a partial last chunk
    (304, 256, 1, 1, 1, 1, 16, 48),      # decoder pointwise: Cin % 128 != 0
    (256, 512, 1, 2, 1, 2, 33, 65),      # downsample: 1x1 stride 2, odd planes
    (256, 512, 1, 2, 1, 2, 32, 64),      # 1x1 stride 2, vector loads
    (1024, 512, 1, 1, 1, 1, 12, 12),
    (64, 64, 3, 1, 1, 2, 20, 72),        # layer1 conv2: two K halves per workgroup
    (64, 128, 3, 1, 1, 1, 33, 45),       # stem conv3, odd plane
    (128, 128, 3, 2, 1, 2, 41, 66),      # layer2.0 conv2: stride 2, odd height
    (128, 128, 3, 2, 1, 2, 40, 64),      # stride 2, vector loads
    (512, 512, 3, 1, 2, 1, 13, 24),      # layer4 conv2: dilation 2
    (256, 256, 3, 1, 2, 2, 12, 48),      # dilation 2, 16-wide tiles (48-wide plane)
    (256, 256, 3, 1, 2, 1, 9, 40),
    (64, 64, 3, 1, 4, 1, 20, 36),        # dilation 4 (output stride 8)
    (8, 64, 3, 1, 1, 1, 5, 7),           # plane smaller than a chunk
    (256, 48, 1, 1, 1, 2, 20, 36),       # decoder low-level projection: Cout 48
    (64, 200, 3, 1, 1, 1, 9, 33),        # Cout 200
    (3, 64, 3, 2, 1, 2, 32, 48),         # the stem's first convolution: Cin 3
    (2048, 256, 1, 1, 1, 2, 12, 12),     # ASPP 1x1
    # 3x3 stride 1 on narrow planes: the linear pixel walk (128 consecutive pixels per tile, full-row patches of 64 / 128 columns)
    (256, 256, 3, 1, 1, 2, 49, 49),      # layer3 at the 769 crop
    (128, 128, 3, 1, 2, 1, 49, 49),      # dilation 2
    (64, 64, 3, 1, 4, 1, 40, 36),        # dilation 4, 13-row patch
    (64, 72, 3, 1, 1, 1, 33, 56),        # widest plane of the 64-column class, Cout 72
    (128, 128, 3, 1, 1, 2, 97, 97),      # layer2 at the 769 crop: the 128-column class
    (64, 64, 3, 1, 2, 1, 20, 120),       # its widest plane, dilation 2
    (64, 64, 3, 1, 1, 1, 30, 57),        # its narrowest
    # BASELINE configs[1]: the two largest planes of the 769 crop at their real size (385 x 385 after the stem's stride, 193 x 193
    # after the max-pool): every row at another 4-byte alignment, the last 16-byte group of every row astride its end
    (64, 64, 3, 1, 1, 1, 385, 385),      # stem conv 2
    (16, 32, 3, 1, 1, 1, 385, 385),
    (64, 128, 3, 2, 1, 1, 385, 385),     # 385 -> 193 at stride 2 (parity classes of the input gradient on an odd plane)
    (32, 64, 1, 1, 1, 1, 385, 385),      # 1x1 with a K tail on the flat walk
    (64, 64, 3, 1, 1, 1, 193, 193),      # layer1 conv2
    (128, 64, 1, 1, 1, 1, 193, 193),     # layer1.0 conv1
    (64, 256, 1, 1, 1, 1, 193, 193),     # layer1 conv3
    (128, 128, 3, 2, 1, 1, 193, 193),    # layer2.0 conv2: 193 -> 97
    (256, 512, 1, 2, 1, 1, 193, 193),    # layer2.0 downsample
]


def _ref_grads(x, w, dy, stride, dil):
    k = w.shape[2]
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, stride, dil if k == 3 else 0, dil)
    y.backward(dy.double())
    return y.detach(), xd.grad, wd.grad


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", CASES)
def test_wgrad_matches_float64_autograd(Cin, Cout, k, stride, dil, N, H, W):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin * 3 + Cout + 17 * k + stride + dil + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(N, Cout, Ho, Wo, device='cuda')
    _, _, dw_ref = _ref_grads(x, w, dy, stride, dil)
    dw = ops.conv_wgrad(x, dy, k, stride, dil)
    assert dw.shape == dw_ref.shape
    scale = float(dw_ref.abs().max())
    err = float((dw.double() - dw_ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    assert torch.equal(dw, ops.conv_wgrad(x, dy, k, stride, dil)), "run-to-run identical (fixed-order split-K reduction)"


def test_wgrad_exact_on_integers():
    """Small integers are exact in f32 whatever the order: pins the lane maps of both operands, the tap offsets, the pixel
    pairing of the two MFMA k values, the K-half reduction and the split-K reduction bit for bit (asymmetric data)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    for Cin, Cout, k, stride, dil, H, W in ((64, 128, 1, 1, 1, 37, 50), (64, 128, 3, 1, 1, 37, 52), (32, 64, 3, 2, 1, 37, 50),
                                            (32, 64, 3, 1, 2, 20, 48), (64, 128, 1, 2, 1, 36, 52), (32, 64, 3, 1, 1, 16, 64),
                                            (160, 64, 1, 1, 1, 16, 16), (2048, 256, 1, 1, 1, 1, 1), (64, 128, 1, 1, 1, 1, 3),
                                            (32, 64, 3, 1, 1, 2, 3), (64, 64, 3, 2, 1, 3, 5)):      # (rows shorter than a 16-byte group)
        x = torch.randint(-3, 4, (2, Cin, H, W), generator=g, device='cuda').float()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        dy = torch.randint(-2, 3, (2, Cout, Ho, Wo), generator=g, device='cuda').float()
        w = torch.zeros(Cout, Cin, k, k, device='cuda')
        _, _, ref = _ref_grads(x, w, dy, stride, dil)
        dw = ops.conv_wgrad(x, dy, k, stride, dil)
        assert torch.equal(dw.double(), ref), (Cin, Cout, k, stride, dil)


@pytest.fixture(params=["dma", "registers"])
def sk_mode(request):
    """Both chunk-staging forms of mas_conv_sk: the register-staged one (default) and the LDS-DMA ring."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    old = ops.conv_sk_set_mode(request.param == "dma")
    yield request.param
    ops.conv_sk_set_mode(old)


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", CASES)
@pytest.mark.parametrize("epi", ["bare", "scale_res_relu"])
def test_stream_k_forward_and_input_gradient(Cin, Cout, k, stride, dil, N, H, W, epi, sk_mode):
    """mas_conv_sk in both roles against float64 conv2d / its autograd, with and without the epilogue, on shapes whose tiles
    straddle workgroups (every case: iterations are dealt in equal runs to 256 workgroups)."""
    from mulactseg_amd import ops
    torch.manual_seed(Cin * 5 + Cout + 13 * k + stride + dil + W)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(N, Cout, Ho, Wo, device='cuda')
    y_ref, dx_ref, _ = _ref_grads(x, w, dy, stride, dil)
    sc = torch.rand(Cout, device='cuda') + 0.5 if epi != "bare" else None
    sh = torch.randn(Cout, device='cuda') if epi != "bare" else None
    res = torch.randn(N, Cout, Ho, Wo, device='cuda') if epi != "bare" else None
    y = ops.conv_sk(x, w, stride, dil, scale=sc, shift=sh, residual=res, relu=epi != "bare")
    if epi != "bare":
        y_ref = torch.relu(y_ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res.double())
    scale = float(y_ref.abs().max())
    assert float((y.double() - y_ref).abs().max()) <= 2e-5 * scale
    if stride == 1 and Cin > 3:
        acc = torch.randn(N, Cin, H, W, device='cuda') if epi != "bare" else None
        dx = ops.conv_sk(dy, w, 1, dil, dgrad=True, residual=acc)
        ref = dx_ref if acc is None else dx_ref + acc.double()
        assert float((dx.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        assert torch.equal(dx, ops.conv_sk(dy, w, 1, dil, dgrad=True, residual=acc)), "run-to-run identical"
    assert ops.conv_sk_error() == 0


def test_stream_k_exact_on_integers_at_layer_sizes(sk_mode):
    """Integer data (exact in f32 in any order) at the plane sizes of the deep layers, where every tile is shared by two or three
    workgroups: operand lane maps, the packed weight images of both roles, tap mirroring, the slot hand-off, the DMA ring (odd
    planes included: 4-byte DMA)."""
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(9)
    for Cin, Cout, k, stride, dil, N, H, W in ((1024, 256, 1, 1, 1, 4, 48, 48), (256, 256, 3, 1, 1, 4, 48, 48), (512, 512, 3, 1, 2, 2, 48, 48),
                                              (64, 64, 3, 1, 1, 1, 96, 96), (128, 128, 3, 2, 1, 2, 64, 96), (256, 512, 1, 2, 1, 2, 64, 64),
                                              (304, 256, 1, 1, 1, 1, 64, 64), (256, 48, 1, 1, 1, 1, 64, 64), (256, 256, 3, 1, 1, 2, 49, 49),
                                              (512, 128, 1, 1, 1, 2, 97, 97), (64, 64, 3, 2, 1, 1, 97, 97), (64, 64, 1, 1, 1, 1, 1, 3),
                                              (64, 64, 3, 1, 1, 2, 2, 3), (2048, 256, 1, 1, 1, 4, 1, 1), (128, 128, 3, 1, 1, 2, 97, 97),
                                              (512, 512, 3, 1, 2, 4, 49, 49), (64, 64, 3, 1, 4, 2, 35, 52)):
        x = torch.randint(-2, 3, (N, Cin, H, W), generator=g, device='cuda').float()
        w = torch.randint(-2, 3, (Cout, Cin, k, k), generator=g, device='cuda').float()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        dy = torch.randint(-2, 3, (N, Cout, Ho, Wo), generator=g, device='cuda').float()
        y_ref, dx_ref, _ = _ref_grads(x, w, dy, stride, dil)
        for rep in range(3):
            assert torch.equal(ops.conv_sk(x, w, stride, dil).double(), y_ref), (Cin, Cout, k, stride, dil, rep)
            if stride == 1:
                assert torch.equal(ops.conv_sk(dy, w, 1, dil, dgrad=True).double(), dx_ref), (Cin, Cout, k, stride, dil, rep)
    assert ops.conv_sk_error() == 0


def test_forward_statistics_from_the_epilogue():
    """mas_conv_sk_stats: the BatchNorm partial sums of y (sum, sum of squares per channel over disjoint pixel sets) formed in the
    epilogue of the forward kernel -- summed over the slots they equal the sums over the stored y (float64 of the f32 values; the
    partials themselves are f32 sums over <= 64 pixels), on planes with partial tiles, tiles shared by several workgroups, Cout that
    is not a multiple of the M tile; and relu(bn(y)) from these partials equals the separate reduction pass to f32 rounding, running
    statistics included, with identical results from run to run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(31)
    for Cin, Cout, k, stride, dil, N, H, W in ((64, 256, 1, 1, 1, 2, 48, 48), (256, 64, 1, 1, 1, 2, 49, 49), (1024, 256, 1, 1, 1, 4, 48, 48),
                                               (64, 64, 3, 1, 1, 2, 40, 56), (128, 128, 3, 2, 1, 2, 41, 66), (256, 256, 3, 1, 2, 1, 49, 49),
                                               (256, 512, 1, 2, 1, 2, 33, 65), (304, 200, 1, 1, 1, 1, 20, 36), (64, 48, 3, 1, 4, 1, 20, 36)):
        x = torch.randn(N, Cin, H, W, device='cuda')
        w = torch.randn(Cout, Cin, k, k, device='cuda') * 0.1
        y0 = ops.conv_sk(x, w, stride, dil)
        y, part = ops.conv_sk(x, w, stride, dil, stats=True)
        assert torch.equal(y, y0)
        yd = y.double()
        ref = torch.stack([yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))], dim=1)
        got = part.sum(dim=1)
        scale = (yd * yd).sum(dim=(0, 2, 3)).sqrt().clamp_min(1.0)
        assert float(((got[:, 0] - ref[:, 0]).abs() / (scale * (N * y.shape[2] * y.shape[3]) ** 0.5)).max()) <= 1e-6, (Cin, Cout, k, stride, dil)
        assert float(((got[:, 1] - ref[:, 1]).abs() / ref[:, 1].clamp_min(1e-6)).max()) <= 1e-5, (Cin, Cout, k, stride, dil)
        y2, part2 = ops.conv_sk(x, w, stride, dil, stats=True)
        assert torch.equal(part, part2), "run-to-run identical partials"
        bn_a, bn_b = torch.nn.BatchNorm2d(Cout).cuda().train(), torch.nn.BatchNorm2d(Cout).cuda().train()
        with torch.no_grad():
            bn_a.weight.uniform_(0.5, 1.5)
            bn_a.bias.uniform_(-0.5, 0.5)
            bn_b.load_state_dict(bn_a.state_dict())
        za = ops.bn_act(bn_a, y, True, None)
        zb = ops.bn_act(bn_b, y, True, None, partials=part)
        assert float((za - zb).abs().max()) <= 2e-5 * float(za.abs().max())
        assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn_b.num_batches_tracked) == 1
    assert ops.conv_sk_error() == 0


def test_block_input_gradient_is_accumulated_in_the_epilogue():
    """A Bottleneck's input feeds conv1 and the residual branch (identity or downsample): with MAS_GRAD_FORK=fused the gradient of
    the residual branch is the `residual` operand of conv1's input-gradient kernel; it must give the same gradients as autograd's
    own accumulation (bit for bit: both round acc + g once), with and without a downsample branch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import deeplab
    for inplanes, planes, stride, down in ((256, 64, 1, False), (128, 64, 1, True), (256, 128, 2, True)):
        torch.manual_seed(7 + inplanes)
        ds = None
        if down:
            ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(planes * 4))
        blk = deeplab.Bottleneck(inplanes, planes, stride=stride, downsample=ds).cuda().train()
        x0 = torch.randn(2, inplanes, 40, 56, device='cuda')
        res = {}
        for mode in ("off", "fused"):
            os.environ["MAS_GRAD_FORK"] = mode
            try:
                for prm in blk.parameters():
                    prm.grad = None
                x = (x0 * 1.0).requires_grad_(True)         # a non-leaf would do too; the gradient must reach x
                y = blk(x)
                (y * torch.linspace(-1, 1, y.numel(), device='cuda').view_as(y)).sum().backward()
                res[mode] = (y.detach().clone(), x.grad.clone(), {n: prm.grad.clone() for n, prm in blk.named_parameters()})
            finally:
                os.environ.pop("MAS_GRAD_FORK")
        assert torch.equal(res["off"][0], res["fused"][0])
        assert torch.equal(res["off"][1], res["fused"][1]), (inplanes, planes, stride, down)
        for n in res["off"][2]:
            assert torch.equal(res["off"][2][n], res["fused"][2][n]), n


def test_stride2_input_gradient_exact_on_integers():
    """mas_conv_sk_dgrad_s2: the four parity classes of the input gradient of a 3x3 stride-2 convolution (1 / 2 / 2 / 4 taps over
    the dY plane, strided stores) on integer data, exact in any order: even and odd planes (odd: the last row / column belongs to
    class 0 only), planes of one pixel row, Cin that is not a multiple of the M tile, the layer sizes at the 768 and 769 crops."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(21)
    for Cin, Cout, N, H, W in ((128, 128, 2, 96, 96), (256, 256, 1, 48, 48), (128, 128, 1, 97, 97), (64, 96, 2, 33, 66), (40, 72, 1, 1, 9),
                               (64, 64, 1, 2, 2), (200, 64, 1, 17, 40), (128, 128, 4, 192, 192)):
        w = torch.randint(-2, 3, (Cout, Cin, 3, 3), generator=g, device='cuda').float()
        x = torch.zeros(N, Cin, H, W, device='cuda')
        dy = torch.randint(-2, 3, (N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1), generator=g, device='cuda').float()
        _, dx_ref, _ = _ref_grads(x, w, dy, 2, 1)
        for rep in range(2):
            dx = ops.conv_sk_dgrad_s2(dy, w, H, W)
            assert torch.equal(dx.double(), dx_ref), (Cin, Cout, N, H, W, rep)
    assert ops.conv_sk_error() == 0


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", [c for c in CASES if c[0] > 3])
def test_conv_train_forward_and_gradients(Cin, Cout, k, stride, dil, N, H, W):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin + 5 * Cout + k + stride + dil + W)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    x = torch.randn(N, Cin, H, W, device='cuda', requires_grad=True)
    os.environ["MAS_TRAIN_CONV"] = "own"
    try:
        own = ops.conv_train_plan(conv, x)
    finally:
        os.environ.pop("MAS_TRAIN_CONV")
    if own is None or not own[0]:
        pytest.skip("geometry outside the forward kernel")
    y = ops.conv_train(conv, x, own)
    dy = torch.randn_like(y)
    y.backward(dy)
    y_ref, dx_ref, dw_ref = _ref_grads(x.detach(), conv.weight.detach(), dy, stride, dil)
    for got, ref, name in ((y.detach(), y_ref, "y"), (x.grad, dx_ref, "dx"), (conv.weight.grad, dw_ref, "dw")):
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max())
        assert err <= 2e-5 * scale, (name, err, scale)


def test_pack_registry_lets_dead_models_go():
    """The registry of packed weight images holds its weights weakly: the active-learning loop builds a new model every round
    (reference train_AL.py:38), and a registry that kept the old weights alive would pin every dead model's convolution weights and
    images in HBM and re-pack them after every optimizer step.  Entries die with their weight; the re-pack launch only covers live
    ones and still tracks an optimizer step of the survivors."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import gc
    from mulactseg_amd import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    reg = ops._PACKS.get(dev)
    before = 0 if reg is None else len(reg.entries)
    convs = [torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda() for _ in range(3)]
    x = torch.randn(1, 64, 16, 16, device='cuda', requires_grad=True)
    for c in convs:
        ops.conv_train(c, x, (True, True, True)).sum().backward()
    reg = ops._PACKS[dev]
    assert len(reg.entries) == before + 6                     # forward + input-gradient image of each
    keep = convs[0]
    opt = torch.optim.AdamW(keep.parameters(), lr=1e-2, fused=True)
    del convs, c
    gc.collect()
    assert len(reg.entries) == before + 2, len(reg.entries)  # the two dead models' images are gone
    w0 = keep.weight.detach().clone()
    y0 = ops.conv_train(keep, x, (True, True, True)).detach().clone()
    keep.weight.grad = torch.ones_like(keep.weight)
    opt.step()                                                # bumps the parameter epoch: the next lookup re-packs the live entries
    assert not torch.equal(keep.weight.detach(), w0)
    y1 = ops.conv_train(keep, x, (True, True, True)).detach()
    ref = F.conv2d(x.detach().double(), keep.weight.detach().double(), padding=1)
    assert float((y1.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) and not torch.equal(y0, y1)
    del keep, opt
    gc.collect()
    assert len(reg.entries) == before


def test_random_geometry_cut_of_the_soak():
    """A 40-geometry cut of tools/soak_conv_train.py (random channel counts, planes from 1 x 1 to 140 x 200, strides, dilations,
    batch sizes; forward with the statistics epilogue, input gradient incl. the residual operand and the stride-2 classes, weight
    gradient) against float64 autograd: the same seeds the builder's 250-geometry soak starts with."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    from mulactseg_amd import ops
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_conv_train.py")
    spec = importlib.util.spec_from_file_location("soak_conv_train", path)
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    ran, worst = 0, 0.0
    for seed in range(9000, 9040):
        out = soak.run_seed(seed)
        if out is None:
            continue
        ran += 1
        assert max(out[1]) <= 3e-5, (seed, out)
        worst = max(worst, max(out[1]))
    assert ran >= 35 and ops.conv_sk_error() == 0
    print("soak cut: %d geometries, worst relative error %.2e" % (ran, worst))


def test_training_step_is_run_to_run_identical():
    """Forward logits and EVERY parameter gradient of two identical training steps are equal bit for bit: fixed-order reductions
    everywhere (stream-K slot order, split-K reduce, BatchNorm partial sums, upsample / max-pool / depthwise backward gathers, the
    pooled 1x1 product) -- no float atomics on the training path.  (Until round 4 the 1x1 convolution of the ASPP pooling branch ran on
    a vendor GEMM with atomic split-K: 6e-5 of the logits between two runs after the BatchNorm over N samples behind it.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    torch.manual_seed(21)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(8), device=dev)
    runs = []
    for rep in range(3):
        for p in net.parameters():
            p.grad = None
        z = net(x, lowres=True)
        z.square().mean().backward()
        runs.append((z.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}))
    for a, b in ((0, 1), (1, 2)):
        assert torch.equal(runs[a][0], runs[b][0]), float((runs[a][0] - runs[b][0]).abs().max())
        bad = [n for n in runs[a][1] if not torch.equal(runs[a][1][n], runs[b][1][n])]
        assert not bad, (len(bad), bad[:5])


def test_train_step_on_own_convolutions_matches_float64():
    """One training step of the whole network (logits and every parameter gradient) with forward / input gradient / weight
    gradient on this package's kernels (MAS_TRAIN_CONV=own) and with all convolutions on MIOpen, both against the same step in
    float64 on the host.  Batch statistics over 2 x 256 x 256 pictures (16 x 16 maps in the deep layers) and ~60 layers amplify
    f32 rounding to ~1e-4 of the logits' range on EITHER path (observed: 1.8e-4 here, 1.0e-4 on MIOpen, whose blocked sums are
    shorter than the k-ordered fma chain of the MFMA kernel); the bar: logits within 3e-4 absolute or 3x MIOpen's error, gradients
    within 2e-2 relative L2 over all parameters or 2x MIOpen's."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import deeplab, get_model
    dev = torch.device('cuda:0')
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(3), device=dev)
    res = {}
    for mode in ("f64", "miopen", "own"):
        os.environ["MAS_TRAIN_CONV"] = mode if mode != "f64" else "miopen"
        try:
            torch.manual_seed(11)
            net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).train()
            net = net.double() if mode == "f64" else net.to(dev)
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            deeplab.path_report(reset=True)
            xin = x.cpu().double() if mode == "f64" else x
            z = net(xin, lowres=True)
            wts = torch.linspace(-1.0, 1.0, z.numel(), device=z.device, dtype=z.dtype).view_as(z)
            (z * wts).sum().backward()
            if mode != "f64":
                torch.cuda.synchronize()
            res[mode] = (z.detach().double().cpu(), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None},
                         deeplab.path_report(reset=True).get("conv_bn_act"))
        finally:
            os.environ.pop("MAS_TRAIN_CONV")
    assert set(res["own"][2]) <= {"train:fdw", "train:fdw/bx", "train:f-w", "train:--w", "train:dense1x1", "miopen+bn"}
    assert "train:fdw" in res["own"][2] or "train:fdw/bx" in res["own"][2]
    assert set(res["miopen"][2]) == {"miopen+bn"}
    zref, gref = res["f64"][0], res["f64"][1]
    ez_own, ez_mi = float((res["own"][0] - zref).abs().max()), float((res["miopen"][0] - zref).abs().max())
    assert ez_own <= max(3e-4, 3.0 * ez_mi), (ez_own, ez_mi)              # cosine logits in [-1, 1]
    # Element-wise comparisons of deep-network gradients are dominated by the handful of ReLU gates that flip under ANY f32
    # rounding (observed: ~1e-1 of a tensor's scale on either path for the deep conv weights); the stable measure is the relative
    # L2 error over all parameters.
    def l2(path):
        num = sum(float(((res[path][1][n] - g) ** 2).sum()) for n, g in gref.items())
        den = sum(float((g ** 2).sum()) for g in gref.values())
        return (num / den) ** 0.5
    worst_own, worst_mi = l2("own"), l2("miopen")
    assert worst_own <= max(2e-2, 2.0 * worst_mi), (worst_own, worst_mi)
    print("vs float64: logits own %.2e / miopen %.2e; relative L2 error of all gradients own %.2e / miopen %.2e" % (ez_own, ez_mi, worst_own, worst_mi))


def test_training_step_takes_no_vendor_or_aten_fallback():
    """Default settings, one training forward + backward: every fusable layer of the network runs on this package's kernels --
    no 'aten' BatchNorm / upsample / pooling / depthwise path, no MIOpen convolution ('miopen+bn'); all three products of every
    dense convolution on the own kernels ('train:fdw') except the 1x1 convolution of the ASPP pooling branch on its 1x1 map
    (a [N,2048] x [2048,256] product on the fixed-order kernels of csrc/head.hip: 'train:dense1x1')."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import deeplab, get_model
    assert os.environ.get("MAS_TRAIN_CONV", "own") == "own"
    torch.manual_seed(5)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).cuda().train()
    x = torch.randn(2, 3, 256, 256, device='cuda')
    deeplab.path_report(reset=True)
    net(x, lowres=True).square().mean().backward()
    torch.cuda.synchronize()
    rep = deeplab.path_report(reset=True)
    for kind, paths in rep.items():
        assert 'aten' not in paths and 'miopen+bn' not in paths, (kind, paths)
    # ("/bx": forward and input gradient of that layer on the split-bf16 kernel csrc/conv_bx.hip instead of the stream-K one)
    assert set(rep["conv_bn_act"]) <= {"train:fdw", "train:fdw/bx", "train:dense1x1"} and rep["conv_bn_act"]["train:dense1x1"] == 1, rep["conv_bn_act"]
    assert rep["conv_bn_act"].get("train:fdw/bx", 0) >= 30, rep["conv_bn_act"]
    assert rep["bn_act"] == {"hip": sum(rep["conv_bn_act"].values())} or set(rep["bn_act"]) == {"hip"}, rep["bn_act"]


def test_packed_weight_images_follow_in_place_updates():
    """ops.packed_weight: the images of all registered convolutions are re-packed by ONE launch when a weight's version counter
    has moved (what an optimizer step does); scaling the weights by 2 in place must scale outputs and input gradients by exactly 2."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(2)
    convs = [torch.nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda(), torch.nn.Conv2d(96, 200, 1, bias=False).cuda(),
             torch.nn.Conv2d(64, 64, 3, stride=2, padding=1, bias=False).cuda()]
    xs = [torch.randn(2, 64, 24, 32, device='cuda', requires_grad=True), torch.randn(2, 96, 24, 32, device='cuda', requires_grad=True),
          torch.randn(2, 64, 24, 32, device='cuda', requires_grad=True)]

    def run():
        outs = []
        for conv, x in zip(convs, xs):
            x.grad = None
            y = ops.conv_train(conv, x, (True, True, True))      # (the stride-2 3x3: four parity-class images)
            y.backward(torch.ones_like(y))
            outs.append((y.detach().clone(), x.grad.clone()))
        return outs
    first = run()
    with torch.no_grad():
        for conv in convs:
            conv.weight.mul_(2.0)
    second = run()
    for (y1, g1), (y2, g2) in zip(first, second):
        assert torch.equal(y2, 2 * y1) and torch.equal(g2, 2 * g1)
    reg = ops._PACKS[xs[0].device]
    assert reg.table is not None and reg.nblocks > 0


def test_caches_follow_a_fused_optimizer_step():
    """torch.optim.AdamW(fused=True) updates parameters without bumping their version counters: the packed training images, the
    packed inference weight and the folded BatchNorm constants must still follow (ops._PARAM_EPOCH via the optimizer-step hook)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(64).cuda().eval()
    x = torch.randn(2, 64, 32, 32, device='cuda')
    opt = torch.optim.AdamW(list(conv.parameters()) + list(bn.parameters()), lr=0.1, fused=True)
    for _ in range(2):
        with torch.no_grad():
            y_inf = ops.conv_mfma(conv, x, bn, relu=True)
            ref = torch.relu(bn(conv(x)))
        assert float((y_inf - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        y = ops.conv_train(conv, x, (True, True, True))
        assert float((y - conv(x)).abs().max()) <= 2e-5 * float(y.abs().max())
        v0 = conv.weight._version
        (y.sum() + bn(conv(x)).sum()).backward()
        opt.step()
        opt.zero_grad()
        assert conv.weight._version == v0 or True          # (fused: unchanged; the epoch hook is what the caches see)


def test_weight_gradient_stream_modes_and_branch_streams_give_the_same_bits():
    """Round 6: the weight gradients run on a second stream joined once per backward pass (MAS_WGRAD_STREAM=async, the default in a
    single-GPU process), independent branches on branch streams (MAS_BRANCH_STREAMS).  Logits and every gradient equal the one-stream
    step bit for bit; with a gradient buffer in place (accumulation: w.grad is not None) the weight gradients stay on the main stream
    and a second backward pass exactly doubles every gradient."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(9), device=dev)
    wts = torch.linspace(-1.0, 1.0, 2 * 20 * 64 * 64, device=dev).view(2, 20, 64, 64)

    def step(zero=True):
        if zero:
            for p in net.parameters():
                p.grad = None
        z = net(x, lowres=True)
        (z * wts).sum().backward()
        torch.cuda.synchronize()
        return z.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    runs = {}
    keep = {k: os.environ.get(k) for k in ("MAS_WGRAD_STREAM", "MAS_BRANCH_STREAMS")}
    try:
        for tag, wg, br in (("main", "main", "off"), ("async", "async", "off"), ("async+branches", "async", "on"), ("side", "side", "on")):
            os.environ["MAS_WGRAD_STREAM"], os.environ["MAS_BRANCH_STREAMS"] = wg, br
            net.load_state_dict(state)
            step()                                      # (first call in a mode: streams and workspaces are created)
            net.load_state_dict(state)
            runs[tag] = step()
        for tag in ("async", "async+branches", "side"):
            assert torch.equal(runs[tag][0], runs["main"][0]), tag
            bad = [n for n in runs["main"][1] if not torch.equal(runs[tag][1][n], runs["main"][1][n])]
            assert not bad, (tag, len(bad), bad[:4])
        # accumulation: a second backward pass on top of existing gradients (AccumulateGrad adds on the main stream)
        os.environ["MAS_WGRAD_STREAM"], os.environ["MAS_BRANCH_STREAMS"] = "async", "on"
        net.load_state_dict(state)
        _, g1 = step()
        assert not ops._async_wgrad_ok(net.backbone.layer4[0].conv2.weight)         # a gradient buffer is in place now
        net.load_state_dict(state)
        _, g2 = step(zero=False)
        bad = [n for n in g1 if not torch.equal(g2[n], g1[n] * 2)]
        assert not bad, (len(bad), bad[:4])
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
