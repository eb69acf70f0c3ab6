#!/usr/bin/env python
"""Intrinsic error of convolution-operand splits, without any kernel: a float64 forward of the model in which every dense convolution
sees operands reconstructed from a split -- (a) three bf16 terms with the six products k_conv_bx keeps, (b) two fp16 terms (power-of-two
scale per weight row and per activation tensor) with the three products hh + hl + lh -- against the unperturbed float64 forward.  The
f32 accumulation of a real kernel comes on top of both in the same way; this isolates what the SPLIT costs.  CPU, eval mode.

    python tools/split_numerics_probe.py [--size 257 321] [--batch 2]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def terms_bf16(x):
    """x (float64 holding float32 values) -> (h, m, l) as float64, each exactly a bf16 value."""
    x32 = x.float()
    h = x32.bfloat16().float()
    r = x32 - h
    m = r.bfloat16().float()
    l = (r - m).bfloat16().float()
    return h.double(), m.double(), l.double()


def terms_fp16(x, dim=None):
    """x -> (h, l, scale): h + l ~ x * scale with h, l exactly fp16 values; scale a power of two per tensor (dim None) or per slice of dim 0."""
    x32 = x.float()
    amax = x32.abs().amax() if dim is None else x32.abs().flatten(1).amax(dim=1).view(-1, *([1] * (x32.dim() - 1)))
    amax = torch.clamp(amax, min=1e-30)
    s = torch.exp2(14.0 - torch.ceil(torch.log2(amax)))          # the largest element lands in (2^13, 2^14]
    xs = x32 * s
    h = xs.half().float()
    l = (xs - h).half().float()
    return h.double(), l.double(), s.double()


def conv_split(mode, x, w, stride, padding, dilation, conv2d):
    conv = lambda a, b: conv2d(a, b, None, stride, padding, dilation)
    if mode == "bf16x3":
        xh, xm, xl = terms_bf16(x)
        wh, wm, wl = terms_bf16(w)
        return conv(xh, wh) + conv(xh, wm) + conv(xm, wh) + conv(xm, wm) + conv(xh, wl) + conv(xl, wh)
    xh, xl, sx = terms_fp16(x)
    wh, wl, sw = terms_fp16(w, dim=0)
    y = conv(xh, wh) + conv(xh, wl) + conv(xl, wh)
    return y / (sx * sw.view(1, -1, 1, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[257, 321])
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    from mulactseg_amd.models import get_model
    torch.manual_seed(7)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).eval()
    # BatchNorm statistics that look like a trained network's (random running stats keep activations O(1) through the depth)
    x32 = torch.randn((a.batch, 3, a.size[0], a.size[1]))
    with torch.no_grad():
        net.train()
        for _ in range(2):
            net(x32)
        net.eval()
    net64 = net.double()
    x = x32.double()
    mode = {"m": None}
    real = F.conv2d

    def patched(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if mode["m"] is None or groups != 1 or weight.shape[1] < 8:
            return real(inp, weight, bias, stride, padding, dilation, groups)
        y = conv_split(mode["m"], inp, weight, stride, padding, dilation, real)
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    torch.nn.functional.conv2d = patched
    import torch.nn.modules.conv as C
    C.F.conv2d = patched
    try:
        with torch.no_grad():
            ref = net64(x, lowres=True)
            out = {}
            for m in ("bf16x3", "fp16x2"):
                mode["m"] = m
                out[m] = net64(x, lowres=True)
            mode["m"] = None
            f32 = net.float()(x32, lowres=True).double()
    finally:
        torch.nn.functional.conv2d = real
        C.F.conv2d = real
    print("cosine logits in [-1, 1], %s, float64 forward as the reference" % (tuple(ref.shape),))
    for name, y in (("plain f32 forward (ATen, this host)", f32), ("three bf16 terms, six products (k_conv_bx's split), exact accumulation", out["bf16x3"]),
                    ("two fp16 terms, three products, exact accumulation", out["fp16x2"])):
        d = (y - ref).abs()
        print("%-78s max |err| %.3e   rms %.3e" % (name, float(d.max()), float(d.pow(2).mean().sqrt())))


if __name__ == "__main__":
    main()
