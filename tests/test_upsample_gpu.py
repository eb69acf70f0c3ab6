"""csrc/upsample.hip against F.interpolate(bilinear, align_corners=False) computed on the CPU (float32 forward within
rounding; backward against the float64 adjoint), determinism of the backward, exact adjointness."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(2, 5, 12, 12, 48, 48), (1, 20, 48, 96, 192, 384), (2, 3, 9, 13, 33, 37), (1, 4, 7, 5, 7, 5), (1, 2, 16, 16, 40, 90)]


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    return ops


@pytest.mark.parametrize("N,C,Hi,Wi,Ho,Wo", CASES)
def test_forward_and_backward(N, C, Hi, Wi, Ho, Wo):
    ops = _gpu()
    g = torch.Generator().manual_seed(Hi * 100 + Wo)
    x = torch.randn((N, C, Hi, Wi), generator=g)
    go = torch.randn((N, C, Ho, Wo), generator=g)
    xd = x.cuda().requires_grad_(True)
    assert ops.upsample_bilinear_supported(xd, (Ho, Wo))
    y = ops.upsample_bilinear(xd, (Ho, Wo))
    ref = F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False)
    assert float((y.detach().cpu() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    y.backward(go.cuda())
    xr = x.double().requires_grad_(True)
    F.interpolate(xr, size=(Ho, Wo), mode='bilinear', align_corners=False).backward(go.double())
    assert float((xd.grad.double().cpu() - xr.grad).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
    first = xd.grad.clone()
    xd.grad = None
    ops.upsample_bilinear(xd, (Ho, Wo)).backward(go.cuda())
    assert torch.equal(first, xd.grad)                                   # no atomics: bit-identical from run to run
    # <U x, g> == <x, U^T g> (the gather is the adjoint of this forward, not of an approximation of it)
    lhs = float((y.detach().double() * go.cuda().double()).sum())
    rhs = float((xd.detach().double() * first.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_model_takes_the_hip_upsample_and_keeps_golden_parity():
    _gpu()
    from test_model import _check, _load
    g, net, x = _load()
    _check(g, net.cuda(), x.cuda(), 1e-4)
