import ctypes, os, sys
import numpy as np
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MAS_CONV_STAMPS"] = "1"
from mulactseg_amd import ops, _lib
cin, cout, k, s, d, N, H, W = [int(v) for v in sys.argv[1:9]]
res_on = len(sys.argv) > 9 and sys.argv[9] == "res"
conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).cuda()
bn = nn.BatchNorm2d(cout).cuda().eval()
x = torch.randn(N, cin, H, W, device='cuda')
with torch.no_grad():
    res = torch.randn_like(conv(x)) if res_on else None
    for _ in range(3):
        y = ops.conv_mfma(conv, x, bn, relu=True, residual=res)
    torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nblk = 4096
buf = np.zeros((nblk, 8), dtype=np.int64)
rc = lib.mas_conv_stamps_read(buf.ctypes.data_as(ctypes.c_void_p), nblk)
live = buf[buf[:, 0] > 0]
print("rc", rc, "blocks", len(live))
names = ["loop_total", "epilogue", "stage", "barrier1", "fetch_issue", "mfma", "barrier2"]
for i, n in enumerate(names):
    print("%-12s mean %9.0f  min %9.0f  max %9.0f" % (n, live[:, i].mean(), live[:, i].min(), live[:, i].max()))
t0 = live[:, 7]
print("start spread (cycles): p50 %d p90 %d max %d" % tuple(np.percentile(t0 - t0.min(), [50, 90, 100])))
