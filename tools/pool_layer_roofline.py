#!/usr/bin/env python
"""Every convolution launch of the pool forward ([4,3,1024,2048], eval) timed ALONE on the chip against the larger of its two bounds:
HBM time of its compulsory bytes (input once, output once, residual once, weights; at the measured copy rate 6.3 TB/s) and matrix time of
its multiplications (split-bf16 bound 419 TFLOP/s).  Sorted by the time above the bound -- where the forward's kernel time can still come
from.  (A launch alone runs faster than in the forward, where it starts on a cold L2 behind another layer; the sum of this table is a
lower estimate of the forward.)

    python tools/pool_layer_roofline.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM, MFMA = 6.3e12, 419e12


def main():
    from mulactseg_amd import ops
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).eval()
    x = torch.randn((4, 3, 1024, 2048), device=dev)
    calls = []
    real_bx, real_dual = ops.conv_bx, ops.conv_bx_dual

    def rec_bx(conv, xx, bn=None, relu=False, residual=None):
        calls.append(("bx", conv, tuple(xx.shape), bn, relu, residual is not None, None, None))
        return real_bx(conv, xx, bn, relu, residual)

    def rec_dual(ca, ba, xa, cb, bb, xb, relu=True):
        calls.append(("dual", ca, tuple(xa.shape), ba, relu, False, (cb, bb), tuple(xb.shape)))
        return real_dual(ca, ba, xa, cb, bb, xb, relu)
    ops.conv_bx, ops.conv_bx_dual = rec_bx, rec_dual
    try:
        with torch.no_grad():
            net(x, lowres=True)
    finally:
        ops.conv_bx, ops.conv_bx_dual = real_bx, real_dual
    torch.cuda.synchronize()
    names = {m: n for n, m in net.named_modules()}
    rows = []
    for kind, conv, xs, bn, relu, has_res, second, xs2 in calls:
        N, Cin, H, W = xs
        k, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
        Cout = conv.out_channels
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        xin = torch.randn(xs, device=dev)
        res = torch.randn((N, Cout, Ho, Wo), device=dev) if has_res else None
        flop = 2.0 * k * k * Cin * Cout * N * Ho * Wo
        byt = 4.0 * (N * Cin * H * W + N * Cout * Ho * Wo * (2 if has_res else 1) + k * k * Cin * Cout * 3 / 2)
        if kind == "dual":
            cb, bb = second
            xb = torch.randn(xs2, device=dev)
            flop += 2.0 * cb.in_channels * Cout * N * Ho * Wo
            byt += 4.0 * (xs2[0] * xs2[1] * xs2[2] * xs2[3] + cb.in_channels * Cout * 3 / 2)
            fn = lambda: real_dual(conv, bn, xin, cb, bb, xb, relu)
        else:
            fn = lambda: real_bx(conv, xin, bn, relu, res)
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        t_h, t_m = byt / HBM * 1e6, flop / MFMA * 1e6
        rows.append((us - max(t_h, t_m), us, t_h, t_m, names.get(conv, "?"), "%dx%d s%d d%d %d->%d @%dx%d%s%s" % (k, k, s, d, Cin, Cout, H, W, " +res" if has_res else "", " +dual" if kind == "dual" else "")))
    rows.sort(reverse=True)
    tot = sum(r[1] for r in rows)
    bound = sum(max(r[2], r[3]) for r in rows)
    print("%d convolution launches on k_conv_bx, alone on the chip: %.2f ms in total; sum of the per-launch bounds %.2f ms (%.2f of it)" % (len(rows), tot / 1e3, bound / 1e3, bound / tot))
    print("%-44s %-42s %8s %8s %8s %8s %6s" % ("layer", "shape", "us", "HBM us", "MFMA us", "above", "frac"))
    for above, us, t_h, t_m, name, shape in rows:
        print("%-44s %-42s %8.1f %8.1f %8.1f %8.1f %6.2f" % (name[:44], shape, us, t_h, t_m, above, max(t_h, t_m) / us))


if __name__ == "__main__":
    main()
