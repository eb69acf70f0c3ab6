"""csrc/conv1x1.hip: small-K 1x1 convolution, plain and with the inference BatchNorm + ReLU + residual epilogue,
against float64 PyTorch; the cached constants follow parameter updates."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K,M,H,W,res", [(2, 128, 64, 24, 40, False), (1, 256, 64, 17, 12, True), (3, 8, 32, 6, 6, False)])
def test_conv1x1_bn_relu_matches_pytorch(N, K, M, H, W, res):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    g = torch.Generator().manual_seed(K + M)
    conv = nn.Conv2d(K, M, 1, bias=False)
    bn = nn.BatchNorm2d(M)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(M, generator=g) + 0.5); bn.bias.copy_(torch.randn(M, generator=g))
        bn.running_mean.copy_(torch.randn(M, generator=g)); bn.running_var.copy_(torch.rand(M, generator=g) + 0.3)
    x = torch.randn((N, K, H, W), generator=g)
    r = torch.randn((N, M, H, W), generator=g) if res else None
    conv.eval(); bn.eval()
    with torch.no_grad():
        ref = bn.double()(conv.double()(x.double()))
        if res:
            ref = ref + r.double()
        ref = F.relu(ref)
        conv, bn = conv.float().cuda(), bn.float().cuda()
        xd = x.cuda()
        assert ops.conv1x1_bn_act_supported(conv, bn, xd)
        y = ops.conv1x1_bn_act(conv, bn, xd, True, r.cuda() if res else None)
        assert float((y.double().cpu() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
        # constants are cached on the module and refreshed when a parameter changes
        conv.weight.mul_(0.5)
        y2 = ops.conv1x1_bn_act(conv, bn, xd, False, None)
        ref2 = bn(conv(xd))
        assert float((y2 - ref2).abs().max()) < 1e-4 * max(1.0, float(ref2.abs().max()))
    assert not ops.conv1x1_bn_act_supported(conv, bn.train(), xd)
    with torch.enable_grad():
        assert not ops.conv1x1_bn_act_supported(conv, bn.eval(), xd)


def test_model_parity_with_conv1x1_kernel():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from test_model import _check, _load
    g, net, x = _load()
    _check(g, net.cuda(), x.cuda(), 1e-4)
