#!/usr/bin/env python
"""Run one convolution geometry repeatedly through csrc/conv_mfma.hip (and MIOpen) -- the target of rocprofv3 PMC passes.
  python tools/conv_probe.py Cin Cout k stride dil N H W [reps] [hip|bx|miopen|both]      (bx: csrc/conv_bx.hip)"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops  # noqa: E402

cin, cout, k, s, d, N, H, W = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
which = sys.argv[10] if len(sys.argv) > 10 else "both"
dev = torch.device('cuda:0')
conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
bn = nn.BatchNorm2d(cout).to(dev).eval()
x = torch.randn(N, cin, H, W, device=dev)
with torch.no_grad():
    for name, fn in (("hip", lambda: ops.conv_mfma(conv, x, bn, relu=True)), ("bx", lambda: ops.conv_bx(conv, x, bn, relu=True)),
                     ("miopen", lambda: conv(x))):
        if which not in (name, "both"):
            continue
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
        flop = 2.0 * cin * k * k * cout * N * ((H - 1) // s + 1) * ((W - 1) // s + 1)
        print("%s: %.1f us  %.1f TFLOP/s" % (name, us, flop / us / 1e6))
