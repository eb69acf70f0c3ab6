"""Where does the 16x16x32 form of k_conv_bx differ from conv2d?  (debug helper; MAS_LIB selects the library)"""
import torch, torch.nn as nn, torch.nn.functional as F
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulactseg_amd import ops
torch.manual_seed(0)
for (Cin, Cout, N, H, W) in ((64, 1024, 4, 64, 128), (64, 256, 2, 32, 64), (64, 1024, 1, 64, 128), (64, 1024, 4, 16, 32)):
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
    bn = nn.BatchNorm2d(Cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(N, Cin, H, W, device='cuda')
        res = torch.randn(N, Cout, H, W, device='cuda')
        for name, r in (("bn_relu", None), ("bn_res_relu", res)):
            ref = F.relu(bn(conv(x)) + (r if r is not None else 0))
            y = ops.conv_bx(conv, x, bn, relu=True, residual=r)
            d = (y - ref).abs()
            bad = d > 1e-3
            print(name, (Cin, Cout, N, H, W), "max err", float(d.max()), "bad", int(bad.sum()), "of", bad.numel())
            if bad.any():
                idx = bad.nonzero()
                print("  n:", idx[:, 0].unique().tolist()[:8], " channels (first 24):", idx[:, 1].unique().tolist()[:24])
                pix = (idx[:, 2] * W + idx[:, 3])
                print("  pixels (first 24):", pix.unique().tolist()[:24], " count of distinct pixels", int(pix.unique().numel()))
                print("  first:", idx[0].tolist(), float(y[tuple(idx[0])]), float(ref[tuple(idx[0])]), "res there", float(r[tuple(idx[0])]) if r is not None else None)
