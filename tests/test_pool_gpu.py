"""csrc/pool.hip against torch.nn.MaxPool2d(3, 2, 1) on the CPU: values and gradients exactly (a maximum is exact; each
input receives at most four gradient terms, summed in a fixed order), odd sizes, ties, -inf, determinism."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,H,W", [(2, 3, 16, 20), (1, 4, 17, 23), (2, 2, 1, 5), (1, 8, 96, 192), (1, 1, 2, 2), (1, 3, 385, 385), (2, 2, 33, 41)])
def test_maxpool_matches_pytorch(N, C, H, W):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn((N, C, H, W), generator=g)
    x = (x * 2).round() / 2                       # plenty of exact ties
    x[0, 0, 0, :2] = float('-inf')
    pool = nn.MaxPool2d(3, 2, 1)
    xr = x.clone().requires_grad_(True)
    yr = pool(xr)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go)
    xd = x.cuda().requires_grad_(True)
    assert ops.maxpool3s2_supported(pool, xd)
    yd = ops.maxpool3s2(xd)
    assert torch.equal(yd.detach().cpu(), yr.detach())
    yd.backward(go.cuda())
    assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 1e-6 * max(1.0, float(xr.grad.abs().max()))
    first = xd.grad.clone()
    xd.grad = None
    ops.maxpool3s2(xd).backward(go.cuda())
    assert torch.equal(first, xd.grad)
    assert not ops.maxpool3s2_supported(nn.MaxPool2d(2, 2), xd)
