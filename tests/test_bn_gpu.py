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


def test_last_block_statistics_equal_the_separate_launch_bit_for_bit():
    """Round 5: the workgroup that writes a channel's last partial sums forms the channel's statistics (forward: mean / invstd / running
    statistics / num_batches_tracked; backward: dgamma / dbeta / the two means) instead of a launch of their own.  Same partials, same
    order of summation: every output equals the three-launch form (MAS_BN_LASTBLOCK=off) bit for bit, on aligned and odd planes, with
    and without residual, across repeated calls (the completion counters must come back to zero)."""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    for (N, C, H, W, res) in ((4, 64, 48, 48, False), (2, 256, 49, 49, True), (4, 128, 97, 33, True), (1, 20, 5, 7, False), (4, 2048, 12, 12, False)):
        torch.manual_seed(N * C + H)
        x = torch.randn(N, C, H, W, device='cuda') * 2.0 + 0.3
        r = torch.randn(N, C, H, W, device='cuda') if res else None
        dy = torch.randn(N, C, H, W, device='cuda')
        outs = {}
        for mode in ("off", "on", "on"):
            os.environ["MAS_BN_LASTBLOCK"] = mode
            try:
                bn = torch.nn.BatchNorm2d(C).cuda().train()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-1, 1, C))
                xi = x.clone().requires_grad_(True)
                ri = r.clone().requires_grad_(True) if res else None
                y = ops.bn_act(bn, xi, True, ri)
                y.backward(dy)
                got = [y.detach(), xi.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()]
                if res:
                    got.append(ri.grad)
            finally:
                os.environ.pop("MAS_BN_LASTBLOCK")
            if mode in outs:
                for a, b in zip(outs[mode], got):
                    assert torch.equal(a, b)
            outs[mode] = got
        for a, b in zip(outs["off"], outs["on"]):
            assert torch.equal(a, b), (N, C, H, W)
        assert int(outs["on"][6]) == 1
    key = (torch.device('cuda', torch.cuda.current_device()), torch.cuda.current_stream().cuda_stream)
    assert int(ops._BN_COUNTERS[key].abs().sum()) == 0
