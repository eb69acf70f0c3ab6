#!/usr/bin/env python
"""Randomised soak of csrc/conv_mfma.hip: many supported geometries against conv2d in float64 (the test-suite runs 24 of them;
this runs hundreds).  python tools/soak_conv.py [n] [seed0]"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
worst, bad = 0.0, 0
for seed in range(seed0, seed0 + n):
    rs = np.random.RandomState(seed)
    k = int(rs.choice([1, 3]))
    stride = int(rs.choice([1, 2]))
    dil = 1 if (k == 1 or stride == 2) else int(rs.choice([1, 2]))
    cin = int(rs.choice([8, 16, 24, 40, 64, 72, 128, 256])) if k == 3 else int(rs.choice([16, 32, 48, 64, 96, 160, 304, 512]))
    cout = int(rs.choice([16, 48, 64, 80, 128, 192, 200, 256, 512]))
    N = int(rs.randint(1, 5))
    H, W = int(rs.randint(1, 140)), int(rs.randint(1, 200))
    torch.manual_seed(seed)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, cin, H, W, device='cuda')
        if not ops.conv_mfma_supported(conv, x):
            continue
        use_bn, use_res, relu = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
        res = torch.randn_like(conv(x)) if use_res else None
        ref = conv.double()(x.double())
        if use_bn:
            ref = bn.double()(ref)
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = F.relu(ref)
        conv.float(); bn.float()
        y = ops.conv_mfma(conv, x, bn if use_bn else None, relu=relu, residual=res)
    err = float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    worst = max(worst, err)
    if err > 2e-5 or y.shape != ref.shape:
        bad += 1
        print("MISMATCH", seed, (cin, cout, k, stride, dil, N, H, W), err, flush=True)
print("soak: %d geometries, %d mismatches, worst relative error %.2e" % (n, bad, worst))
