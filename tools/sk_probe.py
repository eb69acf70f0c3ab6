#!/usr/bin/env python
"""One geometry of mas_conv_sk / mas_conv_wgrad, launched repeatedly (for rocprofv3 --pmc passes):
  python tools/sk_probe.py Cin Cout k stride dil N H W reps [fwd|dgrad|wgrad] [dma]"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops  # noqa: E402

cin, cout, k, s, d, n, h, w, reps = [int(v) for v in sys.argv[1:10]]
what = sys.argv[10] if len(sys.argv) > 10 else "fwd"
if len(sys.argv) > 11 and sys.argv[11] == "dma":
    ops.conv_sk_set_mode(True)
x = torch.randn(n, cin, h, w, device='cuda')
wt = torch.randn(cout, cin, k, k, device='cuda')
ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
dy = torch.randn(n, cout, ho, wo, device='cuda')
pf, pd = ops.conv_sk_pack(wt, s, False), (ops.conv_sk_pack(wt, 1, True) if s == 1 else None)
fn = {"fwd": lambda: ops.conv_sk(x, wt, s, d, packed=pf), "dgrad": lambda: ops.conv_sk(dy, wt, 1, d, dgrad=True, packed=pd),
      "wgrad": lambda: ops.conv_wgrad(x, dy, k, s, d)}[what]
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    fn()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print("%s %s: %.1f us, %.1f TFLOP/s" % (what, sys.argv[1:9], us, 2.0 * n * cout * ho * wo * cin * k * k / us / 1e6))
