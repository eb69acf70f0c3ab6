#!/usr/bin/env python
"""A/B of the quarter-resolution scan at the x4 ratio: generic tap reads vs one period per lane (k_single_pass<..., LOWRES, X4>),
pool batch [4,20,256,512] -> [4,1024,2048].   python tools/lowres_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops                                   # noqa: E402
from mulactseg_amd.synth_pool import device_superpixel_maps     # noqa: E402

dev = torch.device('cuda:0')
B, C, H, W, S = 4, 20, 1024, 2048, 2048
g = torch.Generator(device=dev).manual_seed(3)
zq = (0.5 * torch.randn((B, C, H // 4, W // 4), generator=g, device=dev)).clamp_(-1, 1)
spx = device_superpixel_maps([100 + i for i in range(B)], H, W, S, dev, torch.int16).to(torch.int64)
invT = ops.inv_temperature(0.1)
p = torch.zeros((B, C), dtype=torch.int64, device=dev)
c = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
h = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
for mode in (True, False, True, False):
    for _ in range(50):
        ops.single_pass_accum_lowres(zq, (H, W), spx, S, invT, prob_sum=p, class_sum=c, hist=h, generic=mode)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(200):
        ops.single_pass_accum_lowres(zq, (H, W), spx, S, invT, prob_sum=p, class_sum=c, hist=h, generic=mode)
    b.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per pool batch" % ("generic tap reads" if mode else "x4 period per lane", a.elapsed_time(b) * 1e3 / 200))
