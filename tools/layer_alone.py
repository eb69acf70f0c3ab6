#!/usr/bin/env python
"""One convolution layer of the pool forward alone, N launches (for rocprofv3 --pmc on a single layer):
    python tools/layer_alone.py --cin 64 --cout 128 --h 512 --w 1024 [--k 3 --dil 1 --stride 1 --reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    for n, d in (("cin", 64), ("cout", 128), ("h", 512), ("w", 1024), ("k", 3), ("dil", 1), ("stride", 1), ("reps", 20), ("n", 4)):
        ap.add_argument("--" + n, type=int, default=d)
    a = ap.parse_args()
    from mulactseg_amd import ops
    dev = torch.device('cuda:0')
    conv = torch.nn.Conv2d(a.cin, a.cout, a.k, stride=a.stride, padding=a.dil if a.k == 3 else 0, dilation=a.dil, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(a.cout).to(dev).eval()
    x = torch.randn((a.n, a.cin, a.h, a.w), device=dev)
    with torch.no_grad():
        for _ in range(3):
            ops.conv_bx(conv, x, bn, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.conv_bx(conv, x, bn, True)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    flop = 2.0 * a.k * a.k * a.cin * a.cout * a.n * (a.h // a.stride) * (a.w // a.stride)
    print("%dx%d %d->%d @%dx%d: %.1f us per launch, %.0f TFLOP/s" % (a.k, a.k, a.cin, a.cout, a.h, a.w, us, flop / us / 1e6))


if __name__ == "__main__":
    main()
