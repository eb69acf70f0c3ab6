#!/usr/bin/env python
"""Per-layer A/B of the inference convolutions on one MI355X: csrc/conv_mfma.hip (f32 matrix cores) against csrc/conv_bx.hip (bf16
matrix cores, f32 operands split into three terms), both with the BatchNorm + ReLU epilogue, on the layer shapes of one eval
forward; error of both against float64.

  python tools/bx_table.py [--shape pool|train] [--out gpurun_out/bx_table.md]
"""
import argparse
import collections
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops                        # noqa: E402
from mulactseg_amd.models import get_model           # noqa: E402
from conv_table import timeit                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="pool")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    N, H, W = {"pool": (4, 1024, 2048), "train": (4, 768, 768)}[args.shape]
    dev = torch.device('cuda:0')
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).eval()
    shapes = collections.OrderedDict()

    def hook(name):
        def fn(mod, inp, out):
            x = inp[0]
            key = (mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], mod.dilation[0], mod.groups, tuple(x.shape))
            shapes.setdefault(key, []).append(name)
        return fn
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            m.register_forward_hook(hook(name))
    net.train()
    with torch.no_grad():
        net(torch.randn(N, 3, H, W, device=dev))
    net.eval()
    lines = ["# inference convolutions, batch [%d,3,%d,%d]: f32-MFMA kernel vs split-bf16 kernel (tools/bx_table.py)" % (N, H, W), "",
             "| x | Cin | Cout | k | s | d | H | W | GFLOP | MB | f32 us | TF/s | bx us | TF/s (f32-equivalent) | GB/s | err f32 | err bx | first layer |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = collections.Counter()
    for (cin, cout, k, s, d, g, xs), names in shapes.items():
        if g != 1:
            continue
        x = torch.randn(xs, device=dev)
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        if not ops.conv_mfma_supported(conv, x):
            continue
        with torch.no_grad():
            y32 = ops.conv_mfma(conv, x, bn, relu=True)
            flop = 2.0 * cin * k * k * y32.numel()
            byts = 4.0 * (x.numel() + y32.numel())
            t32 = timeit(lambda: ops.conv_mfma(conv, x, bn, relu=True))
            tbx = ebx = None
            sl = (slice(0, 1), slice(None), slice(0, min(64, y32.shape[2])), slice(None))
            ref = torch.relu(bn.double()(conv.double()(x[0:1, :, :min(64, y32.shape[2]) * s + 2 * d + 2].double())))[:, :, :min(64, y32.shape[2])]
            conv.float(); bn.float()
            sc = float(ref.abs().max())
            # rows near the lower cut see the zero padding in `ref` but real rows in the kernels: compare the upper part only
            hh = max(1, min(64, y32.shape[2]) - 2 * d - 2)
            e32 = float((y32[sl][:, :, :hh].double() - ref[:, :, :hh]).abs().max()) / sc
            if ops.conv_bx_supported(conv, x):
                ybx = ops.conv_bx(conv, x, bn, relu=True)
                ebx = float((ybx[sl][:, :, :hh].double() - ref[:, :, :hh]).abs().max()) / sc
                tbx = timeit(lambda: ops.conv_bx(conv, x, bn, relu=True))
        mult = len(names)
        tot['f32'] += mult * t32
        tot['bx'] += mult * (tbx if tbx is not None else t32)
        tot['best'] += mult * (min(tbx, t32) if tbx is not None else t32)
        tot['flop'] += mult * flop
        lines.append("| %d | %d | %d | %d | %d | %d | %d | %d | %.2f | %.1f | %.0f | %.0f | %s | %s | %s | %.1e | %s | %s |" % (
            mult, cin, cout, k, s, d, xs[2], xs[3], flop / 1e9, byts / 1e6, t32, flop / t32 / 1e6,
            "%.0f" % tbx if tbx else "-", "%.0f" % (flop / tbx / 1e6) if tbx else "-", "%.0f" % (byts / tbx / 1e3) if tbx else "-",
            e32, "%.1e" % ebx if ebx is not None else "-", names[0]))
    lines += ["", "totals per forward (us): f32 kernel %.0f; split-bf16 kernel where supported %.0f; the faster of the two per layer %.0f; %.1f GFLOP"
              % (tot['f32'], tot['bx'], tot['best'], tot['flop'] / 1e9)]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
