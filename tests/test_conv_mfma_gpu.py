"""csrc/conv_mfma.hip: the f32-MFMA implicit-GEMM convolution against torch's conv2d on every layer geometry of the
network (models/segmentation/backbone/resnet.py:129-160, deeplabv3.py:85-137): 1x1 / 3x3, stride 1 / 2, dilation 1 / 2,
odd planes (the 769-crop sizes 385 / 193 / 97 / 49 scaled down), partial tiles, both tile widths, the fused inference
BatchNorm + residual + ReLU epilogue.  v_mfma_f32_32x32x2_f32 is an exact-f32 fma chain in k order, so the tolerance is
that of a differently ordered f32 sum: 2e-5 of the output scale (observed ~2e-6)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Cin, Cout, k, stride, dil, N, H, W
    (128, 64, 1, 1, 1, 2, 24, 40),       # layer1 conv1 (Cout 64 tile, vector loads)
    (64, 256, 1, 1, 1, 1, 25, 33),       # odd plane -> scalar loads, partial tiles
    (304, 256, 1, 1, 1, 1, 16, 48),      # decoder pointwise: Cin % 32 != 0 -> 16-channel chunks
    (256, 512, 1, 2, 1, 2, 33, 65),      # downsample: 1x1 stride 2
    (1024, 512, 1, 1, 1, 1, 12, 12),     # 16-wide tiles
    (64, 64, 3, 1, 1, 2, 20, 70),        # layer1 conv2
    (64, 128, 3, 1, 1, 1, 33, 45),       # stem conv3, odd plane
    (128, 128, 3, 2, 1, 2, 41, 66),      # layer2.0 conv2: stride 2, odd height
    (512, 512, 3, 1, 2, 1, 13, 24),      # layer4 conv2: dilation 2, 16-wide tiles
    (256, 256, 3, 1, 2, 1, 9, 40),       # dilation 2, 32-wide tiles
    (8, 64, 3, 1, 1, 1, 5, 7),           # one chunk, plane smaller than a tile
    (256, 48, 1, 1, 1, 2, 20, 36),       # decoder low-level projection: Cout padded to 64 inside, 48 rows stored
    (64, 200, 3, 1, 1, 1, 9, 33),        # Cout 200 -> 256: the last M tile is half padding
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
def test_conv_mfma_matches_conv2d(Cin, Cout, k, stride, dil, N, H, W, epi):
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
    assert ops.conv_mfma_supported(conv, x)
    with torch.no_grad():
        res = torch.randn_like(conv(x)) if epi == "bn_res_relu" else None
        ref = _ref(x.double(), conv.double(), bn.double() if bn is not None else None, epi != "bare", res.double() if res is not None else None)
        conv.float()
        if bn is not None:
            bn.float()
        y = ops.conv_mfma(conv, x, bn, relu=epi != "bare", residual=res)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)


def test_conv_mfma_exact_on_integers():
    """Small integers are exact in f32 whatever the summation order: A/B lane maps, tap offsets, the accumulator layout
    of the epilogue and the chunk order are all pinned bit for bit (asymmetric data)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    for k, stride, dil in ((1, 1, 1), (3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 2, 1)):
        conv = nn.Conv2d(64, 128, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
        with torch.no_grad():
            conv.weight.copy_(torch.randint(-3, 4, conv.weight.shape, generator=g, device='cuda').float())
            x = torch.randint(-4, 5, (2, 64, 37, 50), generator=g, device='cuda').float()
            y = ops.conv_mfma(conv, x)
            ref = F.conv2d(x.double(), conv.weight.double(), None, stride, dil if k == 3 else 0, dil).float()
            assert torch.equal(y, ref), (k, stride, dil)


def test_packed_weight_cache_follows_the_parameter():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    conv = nn.Conv2d(64, 64, 1, bias=False).cuda()
    x = torch.randn(1, 64, 8, 32, device='cuda')
    with torch.no_grad():
        a = ops.conv_mfma(conv, x)
        conv.weight.mul_(2.0)
        b = ops.conv_mfma(conv, x)
    assert torch.allclose(b, 2 * a, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,H,W,Cout,relu", [(2, 64, 96, 64, True), (1, 37, 40, 64, True), (1, 1024, 2048, 64, True), (3, 16, 8, 32, False)])
def test_stem_conv_matches_conv2d(N, H, W, Cout, relu):
    """csrc/stem.hip (3 -> C, 3x3, stride 2, padding 1, BatchNorm + ReLU epilogue) against conv2d + BatchNorm in float64; odd
    heights, the left / right / top / bottom borders, the full Cityscapes picture."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(H + W)
    conv = nn.Conv2d(3, Cout, 3, stride=2, padding=1, bias=False).cuda()
    bn = nn.BatchNorm2d(Cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, 3, H, W, device='cuda')
        assert ops.stem_conv_supported(conv, x)
        y = ops.stem_conv(conv, x, bn, relu)
        ref = _ref(x.double(), conv.double(), bn.double(), relu, None)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("seed", range(24))
def test_conv_mfma_randomised_geometries(seed):
    """Random supported geometries (channel counts incl. non-multiples of 64 on the output side, odd planes, planes smaller than a
    tile, every stride / dilation combination, all epilogues) against conv2d in float64: the kernel's address arithmetic --
    partial tiles, halo clamps, the chunk walk, the 64-padding of the output channels -- has to hold everywhere
    conv_mfma_supported says yes, not only on the network's own layers."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    rs = np.random.RandomState(1000 + seed)
    k = int(rs.choice([1, 3]))
    stride = int(rs.choice([1, 2]))
    dil = 1 if (k == 1 or stride == 2) else int(rs.choice([1, 2]))
    cin = int(rs.choice([8, 16, 24, 40, 64, 72, 128])) if k == 3 else int(rs.choice([16, 32, 48, 64, 96, 160]))
    cout = int(rs.choice([16, 48, 64, 80, 128, 192, 200]))
    N = int(rs.randint(1, 4))
    H, W = int(rs.randint(3, 70)), int(rs.randint(3, 90))
    torch.manual_seed(seed)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, cin, H, W, device='cuda')
        assert ops.conv_mfma_supported(conv, x), (cin, cout, k, stride, dil)
        use_bn, use_res, relu = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
        res = torch.randn_like(conv(x)) if use_res else None
        ref = _ref(x.double(), conv.double(), bn.double() if use_bn else None, relu, res.double() if use_res else None)
        conv.float(); bn.float()
        y = ops.conv_mfma(conv, x, bn if use_bn else None, relu=relu, residual=res)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (cin, cout, k, stride, dil, N, H, W)
