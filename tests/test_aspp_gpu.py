"""K7 (fused ASPP depthwise triple) against plain PyTorch fp32 conv2d of the same op: forward, input gradient and
weight gradients within fp32 rounding; deterministic weight gradients; model-level parity unchanged."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _need():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("N,C,H,W,dil", [(2, 64, 48, 48, (6, 12, 18)), (1, 32, 64, 128, (6, 12, 18)), (2, 16, 9, 13, (6, 12, 18)), (1, 8, 150, 130, (6, 12, 18)),
                                         (2, 8, 32, 64, (12, 24, 36)),      # output stride 8: the padded-plane kernels with D = 12
                                         (3, 5, 20, 12, (6, 12, 18)),       # a plane narrower than the largest dilation (taps entirely in the padding)
                                         (1, 4, 24, 32, (5, 10, 15)),       # odd dilations: the bounds-tested kernels
                                         (1, 4, 49, 49, (6, 12, 18))])      # the 769 crop's plane: rows that are no whole 16-byte groups
def test_depthwise_triple_matches_conv2d(N, C, H, W, dil):
    """k_dw3_pad / k_dw3_bwd_w_pad (zero-padded LDS plane; W % 4 == 0, dilations D, 2D, 3D with D in {6, 12}) and the bounds-tested
    kernels every other launch takes, against float64 conv2d."""
    _need()
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(N * 100 + C)
    x = torch.randn((N, C, H, W), generator=g, device='cuda', requires_grad=True)
    ws = [torch.randn((C, 1, 3, 3), generator=g, device='cuda', requires_grad=True) for _ in range(3)]
    ys = ops.aspp_depthwise3(x, ws[0], ws[1], ws[2], dil)
    # float64 reference on the CPU (independent of MIOpen's own algorithm choice)
    xr = x.detach().double().cpu().requires_grad_(True)
    wr = [w.detach().double().cpu().requires_grad_(True) for w in ws]
    yr = [F.conv2d(xr, w, padding=d, dilation=d, groups=C) for w, d in zip(wr, dil)]
    for y, r in zip(ys, yr):
        assert float((y.double().cpu() - r).abs().max()) < 1e-5 * max(1.0, float(r.abs().max()))
    go = [torch.randn(y.shape, generator=g, device='cuda') for y in ys]
    torch.autograd.backward(ys, go)
    torch.autograd.backward(yr, [t.double().cpu() for t in go])
    assert float((x.grad.double().cpu() - xr.grad).abs().max()) < 1e-5 * max(1.0, float(xr.grad.abs().max()))
    for w, r in zip(ws, wr):
        assert float((w.grad.double().cpu() - r.grad).abs().max()) < 2e-4 * max(1.0, float(r.grad.abs().max()))
    # determinism of the weight gradient (fixed reduction order)
    first = [w.grad.clone() for w in ws]
    for w in ws:
        w.grad = None
    x.grad = None
    torch.autograd.backward(ops.aspp_depthwise3(x, ws[0], ws[1], ws[2], dil), go)
    assert all(torch.equal(a, w.grad) for a, w in zip(first, ws))


@pytest.mark.parametrize("N,C,H,W,d", [(2, 48, 48, 48, 1), (4, 38, 192, 192, 1), (1, 16, 37, 53, 1), (2, 8, 20, 300, 2), (1, 4, 16, 16, 6), (1, 5, 70, 260, 1),
                                       (2, 3, 5, 8, 1), (2, 6, 193, 193, 1), (1, 4, 33, 18, 1), (1, 3, 20, 259, 1)])
def test_single_depthwise_matches_conv2d(N, C, H, W, d):
    """Decoder depthwise 3x3 (mas_depthwise3x3_*) against float64 conv2d: forward, dx, dw; dw deterministic."""
    _need()
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(7 * N + C + d)
    x = torch.randn((N, C, H, W), generator=g, device='cuda', requires_grad=True)
    w = torch.randn((C, 1, 3, 3), generator=g, device='cuda', requires_grad=True)
    assert ops.depthwise3x3_supported(x, d)
    y = ops.depthwise3x3(x, w, d)
    xr = x.detach().double().cpu().requires_grad_(True)
    wr = w.detach().double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=d, dilation=d, groups=C)
    assert float((y.double().cpu() - yr).abs().max()) < 1e-5 * max(1.0, float(yr.abs().max()))
    go = torch.randn(y.shape, generator=g, device='cuda')
    y.backward(go)
    yr.backward(go.double().cpu())
    assert float((x.grad.double().cpu() - xr.grad).abs().max()) < 1e-5 * max(1.0, float(xr.grad.abs().max()))
    assert float((w.grad.double().cpu() - wr.grad).abs().max()) < 2e-4 * max(1.0, float(wr.grad.abs().max()))
    first = w.grad.clone()
    w.grad = None; x.grad = None
    ops.depthwise3x3(x, w, d).backward(go)
    assert torch.equal(first, w.grad)


def test_separable_conv_module_takes_the_hip_path():
    _need()
    from mulactseg_amd.models.deeplab import AtrousSeparableConvolution
    m = AtrousSeparableConvolution(24, 16, 3, padding=1, dilation=1).cuda()
    x = torch.randn(2, 24, 40, 56, device='cuda')
    ref = m.body(x)
    out = m(x)
    assert float((out - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_model_uses_fused_aspp_and_keeps_parity():
    """The G4 model golden (executed reference) still holds with the K7 path active on the GPU."""
    _need()
    from test_model import _check, _load
    g, net, x = _load()
    net = net.cuda()
    assert net.classifier.aspp._fused_depthwise(torch.zeros(1, 2048, 9, 11, device='cuda')) is not None
    _check(g, net, x.cuda(), 1e-4)
