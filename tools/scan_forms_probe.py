#!/usr/bin/env python
"""Run the two forms of the acquisition scan on the same quarter-resolution logits (target of rocprofv3 PMC passes):
k_single_pass_ring on the upsampled [4,20,1024,2048] tensor, and k_single_pass<LOWRES> on the [4,20,256,512] one."""
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
full = ops.upsample_bilinear(zq, (H, W))
p = torch.zeros((B, C), dtype=torch.int64, device=dev)
c = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
h = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
for _ in range(30):
    ops.single_pass_accum(full, spx, S, invT, prob_sum=p, class_sum=c, hist=h)
for _ in range(30):
    ops.single_pass_accum_lowres(zq, (H, W), spx, S, invT, prob_sum=p, class_sum=c, hist=h)
torch.cuda.synchronize()
print("done")
