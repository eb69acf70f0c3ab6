#!/usr/bin/env python
"""Randomised soak of the split-bf16 kernels (csrc/conv_bx.hip: forward with every epilogue and the input-gradient role;
csrc/conv_wgrad_bx.hip) against float64: any channel counts, planes from 1 x 1 to 140 x 200 (odd ones included), every supported
stride / dilation.  Run without the caching allocator (PYTORCH_NO_CUDA_MEMORY_CACHING=1) an out-of-bounds access faults.
  python tools/soak_conv_bx.py [n] [seed0]"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
lib = _lib.load()
worst, bad, ran = 0.0, 0, [0, 0, 0]
for seed in range(seed0, seed0 + n):
    rs = np.random.RandomState(seed)
    k = int(rs.choice([1, 3]))
    stride = int(rs.choice([1, 2])) if k == 1 else (2 if rs.randint(4) == 0 else 1)       # (a quarter of the 3x3 draws: the strided form)
    dil = 1 if (k == 1 or stride == 2) else int(rs.choice([1, 2]))
    cin = int(rs.choice([3, 8, 16, 20, 24, 40, 64, 72, 128, 256])) if k == 3 else int(rs.choice([16, 32, 48, 50, 64, 96, 160, 304, 512]))
    if k == 3 and stride == 2:
        cin = int(rs.choice([32, 64, 96, 128, 256]))
    cout = int(rs.choice([16, 48, 64, 80, 128, 192, 200, 256, 512]))
    N = int(rs.randint(1, 5))
    H, W = int(rs.randint(1, 140)), int(rs.randint(1, 200))
    if stride == 2:
        H, W = 2 * int(rs.randint(1, 60)), 8 * int(rs.randint(1, 24))
    torch.manual_seed(seed)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, cin, H, W, device='cuda')
        if not ops.conv_bx_supported(conv, x):
            continue
        use_bn, use_res, relu = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
        y0 = conv(x)
        res = torch.randn_like(y0) if use_res else None
        ref = conv.double()(x.double())
        if use_bn:
            ref = bn.double()(ref)
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = F.relu(ref)
        conv.float(); bn.float()
        y = ops.conv_bx(conv, x, bn if use_bn else None, relu=relu, residual=res)
        errs = [float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))]
        ran[0] += 1
        if stride == 1 and lib.mas_conv_bx_supported(k, 1, dil, cout, cin, H, W):          # the input-gradient role
            dy = torch.randn_like(y0)
            g = torch.randn_like(x) if use_res else None
            dref = torch.nn.grad.conv2d_input(x.shape, conv.weight.double(), dy.double(), 1, dil if k == 3 else 0, dil)
            if g is not None:
                dref = dref + g.double()
            dx = ops.conv_bx_raw(dy, conv.weight.detach(), dil, dgrad=True, residual=g)
            errs.append(float((dx.double() - dref).abs().max()) / max(1.0, float(dref.abs().max())))
            ran[1] += 1
            if k == 1 and lib.mas_conv_wgrad_bx_supported(N, cin, H, W, cout):
                wref = torch.einsum('nmp,ncp->mc', dy.double().flatten(2), x.double().flatten(2))
                dw = ops.conv_wgrad_bx(x, dy)[:, :, 0, 0]
                errs.append(float((dw.double() - wref).abs().max()) / max(1.0, float(wref.abs().max())))
                ran[2] += 1
    err = max(errs)
    worst = max(worst, err)
    if err > 2e-5 or y.shape != ref.shape:
        bad += 1
        print("MISMATCH", seed, (cin, cout, k, stride, dil, N, H, W), errs, flush=True)
print("soak: %d forward / %d input-gradient / %d weight-gradient products, %d mismatches, worst relative error %.2e" % (ran[0], ran[1], ran[2], bad, worst))
