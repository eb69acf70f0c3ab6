#!/usr/bin/env python
"""What a presplit (bx3) input buys the consumer kernels, per layer shape: mas_conv_bx_fwd on the f32 tensor against
mas_conv_bx_fwd_pre on its bx3 form (same products, same bits), and what the conversion costs as a pass of its own.
  python tools/bx_pre_table.py [--shape pool|train] [--out gpurun_out/bx_pre_table.md]"""
import argparse
import collections
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops                  # noqa: E402
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
    # (shape walk through the nn.Modules: training mode with MAS_TRAIN_CONV=miopen calls every Conv2d module, so the hooks see every
    #  layer; a quarter-size picture, the planes scaled back by 4 below)
    os.environ["MAS_TRAIN_CONV"] = "miopen"
    net.train()
    with torch.no_grad():
        net(torch.randn(2, 3, H // 4, W // 4, device=dev))
    os.environ.pop("MAS_TRAIN_CONV")
    lib = _lib.load()
    lines = ["# presplit (bx3) input: consumer kernel time per layer, batch [%d,3,%d,%d] (us per call)" % (N, H, W), "",
             "| x | Cin | Cout | k | d | H | W | M tiles | f32 input | bx3 input | gain | split pass (f32 -> bx3) | first layer |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = collections.Counter()
    for (cin, cout, k, s, d, g, xs), names in shapes.items():
        h, w = xs[2] * 4, xs[3] * 4
        if cin * h * w * N * 4 > 3 << 30:
            continue
        if g != 1 or s != 1 or cin < 8 or h * w < 256 or not lib.mas_conv_bx_supported(k, 1, d, cin, cout, h, w):
            continue
        conv = nn.Conv2d(cin, cout, k, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        with torch.no_grad():
            x = torch.randn((N, cin, h, w), device=dev)
            x3 = ops.bx3_split(x)
            a = ops.conv_bx(conv, x, bn, True)
            b = ops.conv_bx_pre(conv, x3, bn, True)
            assert torch.equal(a, b), names[0]
            t_f = timeit(lambda: ops.conv_bx(conv, x, bn, True))
            t_p = timeit(lambda: ops.conv_bx_pre(conv, x3, bn, True))
            t_s = timeit(lambda: ops.bx3_split(x))
        mult = len(names)
        bm = 128 if (k == 1 and cout % 128 == 0) else 64
        tot['f'] += mult * t_f
        tot['p'] += mult * t_p
        lines.append("| %d | %d | %d | %d | %d | %d | %d | %d | %.0f | %.0f | %.0f %% | %.0f | %s |" % (
            mult, cin, cout, k, d, h, w, -(-cout // bm), t_f, t_p, 100.0 * (t_f - t_p) / t_f, t_s, names[0]))
    lines += ["", "per forward (us, the listed stride-1 layers): f32 inputs %.0f, presplit inputs %.0f" % (tot['f'], tot['p'])]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
