#!/usr/bin/env python
"""Per-layer table of the dense convolutions of DeepLabv3+WN / ResNet50-deepstem (models/deeplab.py) on one MI355X:
FLOP, compulsory bytes, time of MIOpen's kernel (+ the separate inference BatchNorm pass it needs) and of the f32-MFMA
implicit-GEMM kernel with the BatchNorm + ReLU epilogue (csrc/conv_mfma.hip), TFLOP/s, and which roofline bounds the layer.

  python tools/conv_table.py [--shape pool|train|train769] [--out gpurun_out/conv_table.md]
"""
import argparse
import collections
import sys
import os

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops                        # noqa: E402
from mulactseg_amd.models import get_model           # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def timeit(fn, n=10):
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
    ap.add_argument("--shape", default="pool")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    N, H, W = {"pool": (4, 1024, 2048), "train": (4, 768, 768), "train769": (4, 769, 769)}[args.shape]
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
    net.train()             # in eval mode most convolutions bypass Module.__call__ (fused MFMA path): collect the shapes in train mode
    with torch.no_grad():
        net(torch.randn(N, 3, H, W, device=dev))
    net.eval()
    rows = []
    tot = collections.Counter()
    for (cin, cout, k, s, d, g, xs), names in shapes.items():
        if g != 1:
            continue
        x = torch.randn(xs, device=dev)
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
        bn = nn.BatchNorm2d(cout).to(dev).eval()
        with torch.no_grad():
            y = conv(x)
            flop = 2.0 * cin * k * k * y.numel()
            byts = 4.0 * (x.numel() + y.numel() + conv.weight.numel())
            t_mi = timeit(lambda: conv(x))
            t_bn = timeit(lambda: ops.bn_act(bn, y, True))
            t_hip = None
            if ops.conv_mfma_supported(conv, x):
                out = ops.conv_mfma(conv, x, bn, relu=True)
                ref = ops.bn_act(bn, y, True)
                err = float((out - ref).abs().max() / ref.abs().max())
                t_hip = timeit(lambda: ops.conv_mfma(conv, x, bn, relu=True))
        floor = max(flop / PEAK_TF / 1e6, byts / PEAK_GBS / 1e3)
        rows.append((len(names), cin, cout, k, s, d, xs[2], xs[3], flop / 1e9, byts / 1e6, t_mi, t_bn, t_hip, floor,
                     "mfma" if flop / PEAK_TF / 1e6 > byts / PEAK_GBS / 1e3 else "hbm", err if t_hip else None, names[0]))
        tot['mi'] += len(names) * (t_mi + t_bn)
        tot['hip'] += len(names) * (t_hip if t_hip is not None and t_hip < t_mi + t_bn else t_mi + t_bn)
        tot['hip_all'] += len(names) * (t_hip if t_hip is not None else t_mi + t_bn)
        tot['floor'] += len(names) * floor
        tot['flop'] += len(names) * flop
    lines = ["# dense convolutions of one eval forward, batch [%d,3,%d,%d] (tools/conv_table.py --shape %s)" % (N, H, W, args.shape), "",
             "f32 peak %.1f TFLOP/s, HBM peak %.0f GB/s; floor = max(FLOP / peak, compulsory bytes / peak).  MIOpen column = its conv kernel(s); "
             "+BN = the separate fused BatchNorm+ReLU pass (csrc/bn.hip) that follows it; MFMA column = csrc/conv_mfma.hip with "
             "the BatchNorm + ReLU in the epilogue." % (PEAK_TF, PEAK_GBS), "",
             "| x | Cin | Cout | k | s | d | H | W | GFLOP | MB | MIOpen us | +BN us | TF/s (conv) | MFMA+BN us | TF/s | floor us | bound | rel err | first layer |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        mult, cin, cout, k, s, d, h, w, gf, mb, t_mi, t_bn, t_hip, floor, bound, err, name = r
        lines.append("| %d | %d | %d | %d | %d | %d | %d | %d | %.2f | %.1f | %.0f | %.0f | %.0f | %s | %s | %.0f | %s | %s | %s |" % (
            mult, cin, cout, k, s, d, h, w, gf, mb, t_mi, t_bn, gf / t_mi * 1e3, "%.0f" % t_hip if t_hip else "-",
            "%.0f" % (gf / t_hip * 1e3) if t_hip else "-", floor, bound, "%.1e" % err if err is not None else "-", name))
    lines += ["", "totals per forward (us): MIOpen conv + BN pass %.0f; MFMA kernel where it wins %.0f; MFMA kernel wherever supported %.0f; "
              "floor %.0f; %.1f GFLOP" % (tot['mi'], tot['hip'], tot['hip_all'], tot['floor'], tot['flop'] / 1e9)]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
