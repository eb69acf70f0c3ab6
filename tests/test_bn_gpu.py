"""csrc/bn.hip (BatchNorm2d + ReLU + residual add, fused) against torch.nn.BatchNorm2d on the CPU in float64:
forward, running statistics, all gradients, inference mode, determinism; model-level parity and a training step."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(4, 8, 24, 32, True, True), (2, 16, 7, 9, True, False), (4, 64, 1, 1, True, False), (3, 5, 33, 37, False, True),
         (2, 12, 96, 96, True, True), (1, 3, 5, 4, False, False),
         # odd planes longer than one reduction chunk (the 769 crop's 193 x 193): aligned-group path, three of four planes misaligned
         (2, 6, 193, 193, True, True), (1, 5, 97, 97, True, False)]


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    return ops


def _ref(bn64, x, res, relu):
    y = bn64(x)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


@pytest.mark.parametrize("N,C,H,W,relu,with_res", CASES)
def test_training_forward_backward_and_running_stats(N, C, H, W, relu, with_res):
    ops = _gpu()
    g = torch.Generator().manual_seed(N * 1000 + C * 10 + H)
    x = torch.randn((N, C, H, W), generator=g) * 2 + 0.5
    res = torch.randn((N, C, H, W), generator=g) if with_res else None
    go = torch.randn((N, C, H, W), generator=g)
    bn = nn.BatchNorm2d(C, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref_bn = copy.deepcopy(bn).double().train()
    dev_bn = copy.deepcopy(bn).cuda().train()
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    yr = _ref(ref_bn, xr, rr, relu)
    yr.backward(go.double())
    xd = x.cuda().requires_grad_(True)
    rd = res.cuda().requires_grad_(True) if with_res else None
    assert ops.bn_act_supported(dev_bn, xd, rd)
    yd = ops.bn_act(dev_bn, xd, relu, rd)
    yd.backward(go.cuda())
    tol = lambda ref: 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((yd.detach().double().cpu() - yr.detach()).abs().max()) < tol(yr.detach())
    assert float((xd.grad.double().cpu() - xr.grad).abs().max()) < 5 * tol(xr.grad)
    if with_res:
        assert float((rd.grad.double().cpu() - rr.grad).abs().max()) < tol(rr.grad)
    assert float((dev_bn.weight.grad.double().cpu() - ref_bn.weight.grad).abs().max()) < 20 * tol(ref_bn.weight.grad)
    assert float((dev_bn.bias.grad.double().cpu() - ref_bn.bias.grad).abs().max()) < 20 * tol(ref_bn.bias.grad)
    assert float((dev_bn.running_mean.double().cpu() - ref_bn.running_mean).abs().max()) < 1e-6
    assert float((dev_bn.running_var.double().cpu() - ref_bn.running_var).abs().max()) < 1e-5
    assert int(dev_bn.num_batches_tracked) == 1
    # determinism: a second identical forward/backward gives the same bits
    dev2 = copy.deepcopy(bn).cuda().train()
    x2 = x.cuda().requires_grad_(True)
    r2 = res.cuda().requires_grad_(True) if with_res else None
    y2 = ops.bn_act(dev2, x2, relu, r2)
    y2.backward(go.cuda())
    assert torch.equal(y2, yd) and torch.equal(x2.grad, xd.grad) and torch.equal(dev2.weight.grad, dev_bn.weight.grad)


@pytest.mark.parametrize("N,C,H,W,relu,with_res", CASES[:4])
def test_inference_uses_running_statistics(N, C, H, W, relu, with_res):
    ops = _gpu()
    g = torch.Generator().manual_seed(C)
    x = torch.randn((N, C, H, W), generator=g)
    res = torch.randn((N, C, H, W), generator=g) if with_res else None
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.2)
    bn.eval()
    with torch.no_grad():
        ref = _ref(copy.deepcopy(bn).double(), x.double(), res.double() if with_res else None, relu)
        dev = copy.deepcopy(bn).cuda()
        assert ops.bn_act_supported(dev, x.cuda(), res.cuda() if with_res else None)
        y = ops.bn_act(dev, x.cuda(), relu, res.cuda() if with_res else None)
    assert float((y.double().cpu() - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
    # a frozen BatchNorm inside a training graph is left to PyTorch
    assert not ops.bn_act_supported(dev, x.cuda().requires_grad_(True))


def test_model_training_step_matches_unfused_modules():
    """One SGD step of a Bottleneck stack with the fused path vs the same modules run op by op."""
    ops = _gpu()
    from mulactseg_amd.models import deeplab
    torch.manual_seed(0)
    blk = deeplab.Bottleneck(32, 8, downsample=None).cuda().train()
    ref = copy.deepcopy(blk)
    x = torch.randn(4, 32, 20, 24, device='cuda')

    def unfused(m, t):
        y = F.relu(m.bn1(m.conv1(t)))
        y = F.relu(m.bn2(m.conv2(y)))
        return F.relu(m.bn3(m.conv3(y)) + t)

    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = blk(xa), unfused(ref, xb)
    assert float((ya - yb).detach().abs().max()) < 1e-4
    ya.square().mean().backward()
    yb.square().mean().backward()
    assert float((xa.grad - xb.grad).abs().max()) < 1e-5 * max(1.0, float(xb.grad.abs().max()))
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) < 2e-4 * max(1.0, float(q.grad.abs().max())), n
    for (n, b), (_, c) in zip(blk.named_buffers(), ref.named_buffers()):
        assert float((b.double() - c.double()).abs().max()) < 1e-5, n


def test_model_golden_parity_with_fused_bn():
    _gpu()
    from test_model import _check, _load
    g, net, x = _load()
    _check(g, net.cuda(), x.cuda(), 1e-4)


def test_tensors_with_a_storage_offset_take_the_element_path_and_agree():
    """x / residual / upstream gradient that are contiguous views at an odd element offset (planes misaligned differently from the
    freshly allocated outputs): the launch falls back to element-wise groups; values agree with those on aligned copies to rounding."""
    ops = _gpu()
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 2, 4, 9, 11
    big = torch.randn((N * C * H * W + 3,), generator=g).cuda()
    bigr = torch.randn((N * C * H * W + 1,), generator=g).cuda()
    x_view = big[3:].view(N, C, H, W)
    r_view = bigr[1:].view(N, C, H, W)
    bn = nn.BatchNorm2d(C).cuda().train()
    outs = []
    for xs, rs in ((x_view, r_view), (x_view.clone(), r_view.clone())):
        b = copy.deepcopy(bn)
        xq = xs.detach().requires_grad_(True)
        rq = rs.detach().requires_grad_(True)
        y = ops.bn_act(b, xq, True, rq)
        y.backward(torch.ones_like(y) * 0.5)
        outs.append((y.detach(), xq.grad, rq.grad, b.weight.grad, b.running_var.clone()))
    for a, c in zip(*outs):         # the partial sums are grouped by address, so the last bits may differ between the two alignments
        assert float((a - c).abs().max()) <= 2e-6 * max(1.0, float(c.abs().max()))


# ---- round 4: the BatchNorm backward's reduction pass folded into the input-gradient kernel of the next convolution ---------------
FUSED_CASES = [
    # N, C (of z), Cout (of the consumer), k, stride, dil, H, W, relu, residual, fork
    (2, 64, 64, 3, 1, 1, 24, 40, True, False, False),       # bn1 -> conv2
    (2, 64, 256, 1, 1, 1, 25, 33, True, False, False),      # bn2 -> conv3, odd plane (mask bits at every alignment)
    (2, 256, 64, 1, 1, 1, 24, 24, True, True, True),        # bn3 (+ identity) -> next block's conv1, the residual branch through the fork
    (1, 128, 128, 3, 2, 1, 41, 66, True, False, False),     # bn1 -> the stride-2 conv2 of layer2.0: four parity-class launches
    (2, 128, 128, 3, 2, 1, 40, 64, True, False, False),
    (1, 200, 72, 3, 1, 2, 49, 49, True, True, False),       # channel tails, the linear pixel walk, dilation 2
    (2, 48, 64, 1, 1, 1, 20, 36, False, False, False),      # a BatchNorm without ReLU (no mask)
    (1, 64, 64, 3, 1, 1, 193, 193, True, False, False),     # layer1 conv2 at the 769 crop
]


@pytest.mark.parametrize("N,C,Cout,k,stride,dil,H,W,relu,with_res,fork", FUSED_CASES)
def test_backward_reduction_in_the_consumers_input_gradient_epilogue(N, C, Cout, k, stride, dil, H, W, relu, with_res, fork):
    """z = relu(bn(u) + r); y = conv(z) (+ a second consumer of z through the fork alias): the gradient of z leaves the convolution's
    input-gradient kernel GATED together with the (sum g, sum g * uhat) partials, the BatchNorm's backward is statistics + one pass
    (ops._BNLink).  All gradients against float64 autograd; the short path must actually have been taken."""
    ops = _gpu()
    torch.manual_seed(N * 100 + C + Cout + k + H)
    bn = nn.BatchNorm2d(C, momentum=0.1).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5)
        bn.bias.copy_(torch.randn(C) * 0.3)
    conv = nn.Conv2d(C, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    u = (torch.randn(N, C, H, W, device='cuda') * 2 + 0.5).requires_grad_(True)
    r = torch.randn(N, C, H, W, device='cuda').requires_grad_(True) if with_res else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    go = torch.randn(N, Cout, Ho, Wo, device='cuda')
    go2 = torch.randn(N, C, H, W, device='cuda')
    # float64 reference
    bn64 = copy.deepcopy(bn).double()
    conv64 = copy.deepcopy(conv).double()
    u64 = u.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if with_res else None
    z64 = _ref(bn64, u64, r64, relu)
    y64 = conv64(z64)
    ((y64 * go.double()).sum() + ((z64 * go2.double()).sum() if fork else 0.0)).backward()
    # the package's path
    z = ops.bn_act(bn, u, relu, r)
    link = getattr(z, '_mas_bn_link', None)
    assert link is not None
    own = ops.conv_train_plan(conv, z)
    assert own is not None and own[1]
    out = ops.conv_train(conv, z, own, fork=fork, bn_link=link)
    if fork:
        y, _, z_alias = out
        ((y * go).sum() + (z_alias * go2).sum()).backward()
    else:
        (out * go).sum().backward()
    assert link.claimed and link.g is None and link.partials is None, "the BatchNorm's backward did not consume the hand-over"
    tol = lambda ref: 3e-5 * max(1.0, float(ref.abs().max()))
    for got, ref, name in ((u.grad, u64.grad, "du"), (bn.weight.grad, bn64.weight.grad, "dgamma"), (bn.bias.grad, bn64.bias.grad, "dbeta"),
                           (conv.weight.grad, conv64.weight.grad, "dw")) + (((r.grad, r64.grad, "dr"),) if with_res else ()):
        err = float((got.double() - ref).abs().max())
        assert err <= (5 if name == "du" else 1) * tol(ref), (name, err, float(ref.abs().max()))
    assert ops.conv_sk_error() == 0


def test_short_path_is_refused_when_the_gradient_is_not_the_handed_over_tensor():
    """A second consumer of z OUTSIDE the convolution (autograd then adds its gradient to the gated one) must send the BatchNorm's
    backward down the long path -- which gates again (idempotent) and reduces itself.  Same answers as float64."""
    ops = _gpu()
    torch.manual_seed(5)
    N, C, H, W = 2, 64, 24, 32
    bn = nn.BatchNorm2d(C).cuda().train()
    conv = nn.Conv2d(C, 64, 3, padding=1, bias=False).cuda()
    u = torch.randn(N, C, H, W, device='cuda').requires_grad_(True)
    go = torch.randn(N, 64, H, W, device='cuda')
    other = torch.randn(N, C, H, W, device='cuda')
    u64 = u.detach().double().requires_grad_(True)
    z64 = F.relu(copy.deepcopy(bn).double()(u64))
    ((copy.deepcopy(conv).double()(z64) * go.double()).sum() + (z64 * other.double()).sum()).backward()
    z = ops.bn_act(bn, u, True)
    link = z._mas_bn_link
    y = ops.conv_train(conv, z, ops.conv_train_plan(conv, z), bn_link=link)
    ((y * go).sum() + (z * other).sum()).backward()             # (z * other): a consumer the convolution does not know about
    assert float((u.grad.double() - u64.grad).abs().max()) <= 1.5e-4 * max(1.0, float(u64.grad.abs().max()))
    assert link.g is None and link.partials is None


def test_whole_step_with_and_without_the_fused_backward_reduction():
    """One training step of the network under MAS_BN_BWD=fused (default) and =separate: same logits bit for bit (the forward does not
    change), gradients equal up to the summation order of the BatchNorm reductions."""
    ops = _gpu()
    import os
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(6), device=dev)
    res = {}
    for mode in ("fused", "separate"):
        os.environ["MAS_BN_BWD"] = mode
        try:
            torch.manual_seed(12)
            net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
            for m in net.modules():
                if isinstance(m, nn.Dropout):
                    m.p = 0.0
            z = net(x, lowres=True)
            wts = torch.linspace(-1.0, 1.0, z.numel(), device=dev).view_as(z)
            (z * wts).sum().backward()
            res[mode] = (z.detach().clone(), {n: p.grad.detach().double() for n, p in net.named_parameters()})
        finally:
            os.environ.pop("MAS_BN_BWD")
    assert torch.equal(res["fused"][0], res["separate"][0])
    num = sum(float(((res["fused"][1][n] - g) ** 2).sum()) for n, g in res["separate"][1].items())
    den = sum(float((g ** 2).sum()) for g in res["separate"][1].values())
    assert (num / den) ** 0.5 <= 1e-4, (num / den) ** 0.5
    assert ops.conv_sk_error() == 0
