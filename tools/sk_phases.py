#!/usr/bin/env python
"""Where one wave of mas_conv_sk spends an iteration (needs a library whose conv_sk.hip was built with -DSK_PHASED -DSK_PHASE_STAMPS: the phased form of the multiply loop, kept for this measurement): cycles of MFMA part 1,
stage + refetch, MFMA part 2, barrier, per wave of workgroup 7.   python tools/sk_phases.py Cin Cout k stride dil N H W"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops  # noqa: E402

cin, cout, k, s, d, n, h, w = [int(v) for v in sys.argv[1:9]]
x = torch.randn(n, cin, h, w, device='cuda')
wt = torch.randn(cout, cin, k, k, device='cuda')
pk = ops.conv_sk_pack(wt, s, False)
for _ in range(5):
    ops.conv_sk(x, wt, s, d, packed=pk)
st = torch.zeros((512 * 4 + 64,), dtype=torch.int64, device='cuda')
ops.conv_sk(x, wt, s, d, packed=pk, stamps=st)
torch.cuda.synchronize()
ph = st.cpu().numpy()[2048:2048 + 64].reshape(8, 8)
print(sys.argv[1:9])
for wv in range(8):
    n_it = max(1, ph[wv, 4])
    print("wave %d: iterations %d; cycles per iteration: mfma part 1 %.0f, stage + refetch %.0f, mfma part 2 %.0f, barrier %.0f, total %.0f; of stage: wait for loads %.0f"
          % (wv, ph[wv, 4], ph[wv, 0] / n_it, ph[wv, 1] / n_it, ph[wv, 2] / n_it, ph[wv, 3] / n_it, ph[wv, :4].sum() / n_it, ph[wv, 5] / n_it))
