"""Training-mode dense convolutions (csrc/conv_wgrad.hip + csrc/conv_mfma.hip through ops._ConvTrain): weight gradient,
input gradient and forward against float64 autograd of torch's conv2d on every layer geometry of the network
(models/segmentation/backbone/resnet.py:129-171, deeplabv3.py:85-137,168-245) -- 1x1 / 3x3, stride 1 / 2, dilation 1 / 2 / 4,
odd planes (the 769-crop sizes scaled down), channel counts that are not multiples of the tiles (3, 48, 304, 200), planes
smaller than one chunk.  v_mfma_f32_32x32x2_f32 is an exact-f32 fma chain, so the tolerance is that of a differently ordered
f32 sum: 2e-5 of the result's scale."""
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
                                            (160, 64, 1, 1, 1, 16, 16)):
        x = torch.randint(-3, 4, (2, Cin, H, W), generator=g, device='cuda').float()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        dy = torch.randint(-2, 3, (2, Cout, Ho, Wo), generator=g, device='cuda').float()
        w = torch.zeros(Cout, Cin, k, k, device='cuda')
        _, _, ref = _ref_grads(x, w, dy, stride, dil)
        dw = ops.conv_wgrad(x, dy, k, stride, dil)
        assert torch.equal(dw.double(), ref), (Cin, Cout, k, stride, dil)


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,N,H,W", [c for c in CASES if c[0] % 16 == 0 and c[1] % 16 == 0])
def test_conv_train_forward_and_gradients(Cin, Cout, k, stride, dil, N, H, W):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    torch.manual_seed(Cin + 5 * Cout + k + stride + dil + W)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    x = torch.randn(N, Cin, H, W, device='cuda', requires_grad=True)
    if not ops.conv_train_supported(conv, x):
        pytest.skip("geometry outside the training path")
    y = ops.conv_train(conv, x)
    dy = torch.randn_like(y)
    y.backward(dy)
    y_ref, dx_ref, dw_ref = _ref_grads(x.detach(), conv.weight.detach(), dy, stride, dil)
    for got, ref, name in ((y.detach(), y_ref, "y"), (x.grad, dx_ref, "dx"), (conv.weight.grad, dw_ref, "dw")):
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max())
        assert err <= 2e-5 * scale, (name, err, scale)
