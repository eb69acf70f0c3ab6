#!/usr/bin/env python
"""Randomised soak of the training convolution kernels (csrc/conv_sk.hip: forward with the statistics epilogue, input gradient at
stride 1 / 2 incl. the residual operand; csrc/conv_wgrad.hip) against float64 autograd of conv2d -- random channel counts, plane
sizes from 1 x 1 to 140 x 200 (aligned and unaligned rows, the linear pixel walk's classes), strides, dilations, batch sizes.
Run it without the caching allocator so that an out-of-bounds read faults:
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/soak_conv_train.py [n] [seed0]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops  # noqa: E402



def geometry(seed):
    """The seed's random geometry (Cin, Cout, k, stride, dil, N, H, W), or None when the draw is degenerate."""
    rs = np.random.RandomState(seed)
    k = int(rs.choice([1, 3]))
    stride = int(rs.choice([1, 1, 2]))
    dil = 1 if (k == 1 or stride == 2) else int(rs.choice([1, 1, 2, 4]))
    cin = int(rs.choice([8, 16, 24, 40, 64, 72, 128, 200, 256, 304]))
    cout = int(rs.choice([16, 48, 64, 80, 128, 192, 200, 256]))
    N = int(rs.randint(1, 4))
    H, W = int(rs.randint(1, 140)), int(rs.randint(1, 200))
    if rs.randint(4) == 0:
        H = W = int(rs.choice([33, 49, 56, 57, 65, 97, 120]))
    if N * cin * H * W < 4 or N * cout * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) < 4:
        return None
    return cin, cout, k, stride, dil, N, H, W


def run_seed(seed):
    """(geometry, relative errors [forward, weight gradient, input gradient?, statistics]) of one random geometry against float64
    autograd, None for a degenerate draw; raises what the kernels raise for a refused geometry."""
    tag = geometry(seed)
    if tag is None:
        return None
    cin, cout, k, stride, dil, N, H, W = tag
    torch.manual_seed(seed)
    x = torch.randn(N, cin, H, W, device='cuda')
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.2
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(N, cout, Ho, Wo, device='cuda')
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yref = F.conv2d(xd, wd, None, stride, dil if k == 3 else 0, dil)
    yref.backward(dy.double())
    y, part = ops.conv_sk(x, w, stride, dil, stats=True)
    res = torch.randn_like(x)
    if stride == 1:
        dx = ops.conv_sk(dy, w, 1, dil, dgrad=True, residual=res) - res
    elif k == 3:
        dx = ops.conv_sk_dgrad_s2(dy, w, H, W)
    else:
        dx = None
    dw = ops.conv_wgrad(x, dy, k, stride, dil)
    errs = [float((y.double() - yref.detach()).abs().max()) / max(1e-6, float(yref.abs().max())),
            float((dw.double() - wd.grad).abs().max()) / max(1e-6, float(wd.grad.abs().max()))]
    if dx is not None:
        errs.append(float((dx.double() - xd.grad).abs().max()) / max(1e-6, float(xd.grad.abs().max()) + 1.0))
    s = part.sum(dim=1)
    yd = y.double()
    errs.append(float((s[:, 0] - yd.sum(dim=(0, 2, 3))).abs().max()) / max(1.0, float(yd.abs().sum(dim=(0, 2, 3)).max())) * 1e-1)
    return tag, errs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    worst, bad, ran = 0.0, 0, 0
    for seed in range(seed0, seed0 + n):
        try:
            out = run_seed(seed)
        except Exception as e:          # a refused geometry is fine; say so
            print("refused", seed, geometry(seed), str(e)[:80], flush=True)
            continue
        if out is None:
            continue
        ran += 1
        e = max(out[1])
        worst = max(worst, e)
        if e > 3e-5:
            bad += 1
            print("MISMATCH", seed, out[0], out[1], flush=True)
    assert ops.conv_sk_error() == 0
    print("soak: %d geometries run, %d mismatches, worst relative error %.2e" % (ran, bad, worst))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
