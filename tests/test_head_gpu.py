"""K8 cosine classifier (csrc/head.hip) against F.normalize + conv2d in float64: logits, feature and proxy gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Ch,K,H,W", [(2, 256, 20, 24, 40), (1, 64, 19, 9, 13), (3, 32, 21, 16, 16)])
def test_cosine_head_matches_pytorch(N, Ch, K, H, W):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator().manual_seed(N * 10 + K)
    f = torch.randn((N, Ch, H, W), generator=g)
    f[0, :, 0, 0] = 0.0                                   # a zero feature vector: logits 0, finite gradients
    proxy = torch.randn((K, Ch, 1, 1), generator=g)
    go = torch.randn((N, K, H, W), generator=g)
    fd, pd = f.cuda().requires_grad_(True), proxy.cuda().requires_grad_(True)
    assert ops.cosine_head_supported(fd, pd)
    out = ops.cosine_head(fd, pd)
    out.backward(go.cuda())
    fr, pr = f.double().requires_grad_(True), proxy.double().requires_grad_(True)
    ref = F.conv2d(F.normalize(fr), F.normalize(pr, dim=1))
    ref.backward(go.double())
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 2e-6
    assert float(out[0, :, 0, 0].abs().max()) == 0.0
    m = torch.ones(N, 1, H, W, dtype=torch.bool); m[0, :, 0, 0] = False      # d/df at |f| = 0 is eps-dominated in the reference
    assert float(((fd.grad.double().cpu() - fr.grad) * m).abs().max()) < 2e-5 * max(1.0, float(fr.grad.abs().max()))
    assert torch.isfinite(fd.grad).all()
    assert float((pd.grad.double().cpu() - pr.grad).abs().max()) < 2e-4 * max(1.0, float(pr.grad.abs().max()))


def test_model_parity_with_cosine_head_kernel():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from test_model import _check, _load
    g, net, x = _load()
    _check(g, net.cuda(), x.cuda(), 1e-4)
