#!/usr/bin/env python
"""Per-workgroup timeline of one mas_conv_sk launch (wall-clock stamps written by the kernel): when do workgroups start, how long
is the pipeline prologue, how even are the end times.   python tools/sk_stamps.py Cin Cout k stride dil N H W [fwd|dgrad]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops  # noqa: E402

cin, cout, k, s, d, n, h, w = [int(v) for v in sys.argv[1:9]]
what = sys.argv[9] if len(sys.argv) > 9 else "fwd"
x = torch.randn(n, cin, h, w, device='cuda')
wt = torch.randn(cout, cin, k, k, device='cuda')
dy = torch.randn(n, cout, (h - 1) // s + 1, (w - 1) // s + 1, device='cuda')
pk = ops.conv_sk_pack(wt, s, what == "dgrad")
fn = (lambda: ops.conv_sk(dy, wt, 1, d, dgrad=True, packed=pk)) if what == "dgrad" else (lambda: ops.conv_sk(x, wt, s, d, packed=pk))
for _ in range(5):
    fn()
st = torch.zeros((512, 4), dtype=torch.int64, device='cuda')
fn0 = fn
fn = (lambda: ops.conv_sk(dy, wt, 1, d, dgrad=True, packed=pk, stamps=st)) if what == "dgrad" else (lambda: ops.conv_sk(x, wt, s, d, packed=pk, stamps=st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record(); fn(); b.record()
torch.cuda.synchronize()
t = st.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t - t0) / 100.0          # 100 MHz
print("%s %s: event time %.1f us, %d workgroups" % (what, sys.argv[1:9], a.elapsed_time(b) * 1e3, len(t)))
for i, name in enumerate(("start", "primed", "loop done", "end")):
    v = us[:, i]
    print("  %-10s min %7.2f  mean %7.2f  max %7.2f us" % (name, v.min(), v.mean(), v.max()))
dur = us[:, 3] - us[:, 0]
print("  workgroup duration: min %.2f mean %.2f max %.2f us;  prologue mean %.2f us;  after-loop (hand-off + epilogue) mean %.2f max %.2f us"
      % (dur.min(), dur.mean(), dur.max(), (us[:, 1] - us[:, 0]).mean(), (us[:, 3] - us[:, 2]).mean(), (us[:, 3] - us[:, 2]).max()))
if os.environ.get("SK_STAMPS_RAW"):
    full = st.cpu().numpy().astype(np.float64)
    P = int((full[:, 0] > 0).sum())
    d = (full[:P, 3] - full[:P, 0]) / 100.0
    s0 = (full[:P, 0] - full[:P, 0].min()) / 100.0
    e = (full[:P, 3] - full[:P, 0].min()) / 100.0
    per = P // 8 if P % 8 == 0 else P
    for x in range(P // per):
        seg = slice(x * per, (x + 1) * per)
        print("  XCD %d: start %.2f..%.2f  end min %.1f mean %.1f max %.1f | first 8 ends: %s | last 8 ends: %s"
              % (x, s0[seg].min(), s0[seg].max(), e[seg].min(), e[seg].mean(), e[seg].max(),
                 " ".join("%.1f" % v for v in e[seg][:8]), " ".join("%.1f" % v for v in e[seg][-8:])))
