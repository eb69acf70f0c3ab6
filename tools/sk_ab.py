#!/usr/bin/env python
"""Totals of the own training convolutions (forward / input gradient on k_conv_sk, weight gradient on k_wgrad) over the layers of
one training step, for A/B runs of two library builds in one GPU session:
    MAS_LIB=/path/to/libA.so python tools/sk_ab.py [--shape train|train769] [--rows]"""
import argparse
import collections
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops                        # noqa: E402
from mulactseg_amd.models import get_model           # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="train")
    ap.add_argument("--rows", action="store_true")
    ap.add_argument("--no-wgrad", action="store_true")
    args = ap.parse_args()
    N, H, W = {"train": (4, 768, 768), "train769": (4, 769, 769)}[args.shape]
    dev = torch.device('cuda:0')
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
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
    with torch.no_grad():
        net(torch.randn(N, 3, H, W, device=dev))
    tot = collections.Counter()
    for (cin, cout, k, s, d, groups, xs), names in shapes.items():
        if groups != 1 or xs[2] * xs[3] == 1 or cin < 8:
            continue
        x = torch.randn(xs, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        Ho, Wo = (xs[2] - 1) // s + 1, (xs[3] - 1) // s + 1
        dy = torch.randn(xs[0], cout, Ho, Wo, device=dev)
        n = len(names)
        gflop = 2.0 * dy.numel() * cin * k * k / 1e9
        pf = ops.conv_sk_pack(w, s, False)
        tf = timeit(lambda: ops.conv_sk(x, w, s, d, packed=pf, stats=True))
        td = 0.0
        if s == 1:
            pd = ops.conv_sk_pack(w, 1, True)
            td = timeit(lambda: ops.conv_sk(dy, w, 1, d, dgrad=True, packed=pd))
        elif k == 3:
            packs = [ops.conv_sk_pack(w, 2, 2 + sub) for sub in range(4)]
            td = timeit(lambda: ops.conv_sk_dgrad_s2(dy, w, xs[2], xs[3], packed=packs))
        tw = 0.0 if args.no_wgrad else timeit(lambda: ops.conv_wgrad(x, dy, k, s, d))
        tot["f"] += n * tf
        tot["d"] += n * td
        tot["w"] += n * tw
        kind = "%dx%d%s" % (k, k, "" if s == 1 else "s2")
        tot["f_" + kind] += n * tf
        tot["d_" + kind] += n * td
        if args.rows:
            print("%d x %4d -> %4d k%d s%d d%d %3dx%-3d %6.2f GF | fwd %6.1f (%5.1f TF) | dgrad %6.1f | wgrad %6.1f" %
                  (n, cin, cout, k, s, d, xs[2], xs[3], gflop, tf, gflop / tf * 1e3, td, tw), flush=True)
    print("lib %s shape %s: per step us: forward %.0f  dgrad %.0f  wgrad %.0f | %s"
          % (os.path.basename(os.environ.get("MAS_LIB", "default")), args.shape, tot["f"], tot["d"], tot["w"],
             "  ".join("%s %.0f" % (k, v) for k, v in sorted(tot.items()) if "_" in k)))
    assert ops.conv_sk_error() == 0


if __name__ == "__main__":
    main()
