#!/usr/bin/env python
"""Where a wave of csrc/conv_bx.hip spends its cycles: run one geometry on the -DBX_STAMPS build (tools/build_variant.sh conv_bx
"-DBX_PF2=0 -DBX_STAMPS" libvar_stamps.so) and print the per-phase cycle sums of wave 0, averaged over the workgroups.
  MAS_LIB=$PWD/mulactseg_amd/libvar_stamps.so python tools/bx_stamps.py Cin Cout k stride dil N H W"""
import ctypes
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops  # noqa: E402

cin, cout, k, s, d, N, H, W = [int(v) for v in sys.argv[1:9]]
dev = torch.device('cuda:0')
conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
bn = nn.BatchNorm2d(cout).to(dev).eval()
x = torch.randn(N, cin, H, W, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(10 * (1 << 17), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        ops.conv_bx(conv, x, bn, relu=True)
    torch.cuda.synchronize()
    lib.mas_conv_bx_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv_bx(conv, x, bn, relu=True)
    b.record()
    torch.cuda.synchronize()
    lib.mas_conv_bx_debug_stamps(None)
st = buf.view(-1, 10).cpu()
st = st[st[:, 8] > 0].double()
names = ["prologue fetch", "stage (wait loads, split, LDS stores)", "barrier 1", "fetch issue", "LDS reads + MFMAs", "barrier 2", "last chunk", "epilogue"]
tot = st[:, 8].mean()
print("%d->%d k%d s%d d%d [%d,%d,%d]: %d workgroups stamped, kernel %.1f us, mean cycles per workgroup %.0f (s_memtime ticks), span of starts %.0f"
      % (cin, cout, k, s, d, N, H, W, st.shape[0], a.elapsed_time(b) * 1e3, tot, float(st[:, 9].max() - st[:, 9].min())))
for i, nm in enumerate(names):
    print("  %-40s %9.0f  %5.1f %%" % (nm, st[:, i].mean(), 100 * st[:, i].mean() / tot))
