#!/usr/bin/env python
"""A handful of launches of each split-bf16 matrix-core kernel at layer shapes of the training step / the pool forward, for counter
runs (tools/pmc_bx_kernels.sh):  python tools/bx_micro.py [--reps 12]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)

    def rnd(*shape):
        return torch.randn(shape, generator=g, device=dev)
    cases = []
    # forward 1x1 (BM 128 x BN 128): a deep layer of the pool forward, a wide one of the training step
    for (n, cin, cout, h, w) in ((4, 1024, 2048, 64, 128), (4, 256, 256, 192, 192)):
        x, wt = rnd(n, cin, h, w), rnd(cout, cin, 1, 1) * 0.05
        pk = ops.conv_bx_pack(wt, 0)
        cases.append(lambda x=x, wt=wt, pk=pk: ops.conv_bx_raw(x, wt, 1, packed=pk, ksplit=1))
    # forward 3x3
    x, wt = rnd(4, 64, 384, 384), rnd(64, 64, 3, 3) * 0.05
    pk = ops.conv_bx_pack(wt, 0)
    cases.append(lambda: ops.conv_bx_raw(x, wt, 1, packed=pk, ksplit=1))
    # weight gradients
    x1, dy1 = rnd(4, 256, 48, 48), rnd(4, 1024, 48, 48)
    cases.append(lambda: ops.conv_wgrad_bx(x1, dy1))
    x2, dy2 = rnd(4, 128, 192, 192), rnd(4, 256, 192, 192)
    cases.append(lambda: ops.conv_wgrad_bx(x2, dy2))
    x3, dy3 = rnd(4, 64, 384, 384), rnd(4, 64, 384, 384)
    cases.append(lambda: ops.conv_wgrad_bx3(x3, dy3, 1))
    x4, dy4 = rnd(4, 512, 48, 48), rnd(4, 512, 48, 48)
    cases.append(lambda: ops.conv_wgrad_bx3(x4, dy4, 2))
    for fn in cases:
        for _ in range(args.reps):
            fn()
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
