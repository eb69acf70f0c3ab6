#!/usr/bin/env python
"""Per-layer table of the dense convolutions of one TRAINING step (batch [4,3,768,768] or 769): forward, input gradient and
weight gradient, MIOpen (through ATen) beside the f32-MFMA kernels of this package (csrc/conv_mfma.hip, csrc/conv_wgrad.hip).

  python tools/conv_train_table.py [--shape train|train769] [--out gpurun_out/conv_train_table.md]
"""
import argparse
import collections
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import ops                        # noqa: E402
from mulactseg_amd.models import get_model           # noqa: E402

PEAK_TF = 157.3


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
    ap.add_argument("--shape", default="train")
    ap.add_argument("--out", default=None)
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
    with torch.no_grad():                              # (no autograd: every convolution goes through its nn.Module call, hooks fire)
        net(torch.randn(N, 3, H, W, device=dev))
    os.environ["MAS_TRAIN_CONV"] = "own"               # the table times every product the kernels support
    rows, tot = [], collections.Counter()
    for (cin, cout, k, s, d, groups, xs), names in shapes.items():
        if groups != 1 or xs[2] * xs[3] == 1:
            continue
        conv = nn.Conv2d(cin, cout, k, stride=s, padding=d if k == 3 else 0, dilation=d, bias=False).to(dev)
        x = torch.randn(xs, device=dev)
        w = conv.weight.detach()
        y = F.conv2d(x, w, None, s, conv.padding, d)
        dy = torch.randn_like(y)
        gflop = 2.0 * y.numel() * cin * k * k / 1e9
        pad = conv.padding

        def mi(mask):
            return lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), pad, (d, d), False, (0, 0), 1, mask)
        t = {"mi_f": timeit(lambda: F.conv2d(x, w, None, s, pad, d)),
             "mi_d": timeit(mi((True, False, False))) if cin > 3 else 0.0,
             "mi_w": timeit(mi((False, True, False)))}
        t["my_w"] = timeit(lambda: ops.conv_wgrad(x, dy, k, s, d))
        own = ops.conv_train_plan(conv, x)
        if own is not None and own[0]:
            pf = ops.conv_sk_pack(w, s, False)
            t["my_f"] = timeit(lambda: ops.conv_sk(x, w, s, d, packed=pf))
            if own[1]:
                pd = ops.conv_sk_pack(w, 1, True)
                t["my_d"] = timeit(lambda: ops.conv_sk(dy, w, 1, d, dgrad=True, packed=pd))
        dw = ops.conv_wgrad(x, dy, k, s, d)
        ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, (s, s), pad, (d, d), False, (0, 0), 1,
                                                  (False, True, False))[1]
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        import ctypes
        plan = (ctypes.c_int * 6)()
        from mulactseg_amd import _lib
        _lib.load().mas_conv_wgrad_plan(xs[0], cin, xs[2], xs[3], cout, k, s, d, plan)
        n = len(names)
        for kk, v in t.items():
            tot[kk] += n * v
        for a, b in (("f", "mi_f"), ("d", "mi_d"), ("w", "mi_w")):
            tot["best_" + a] += n * min(t.get("my_" + a, 1e30), t[b]) if t[b] > 0 else 0
        tot["gflop"] += n * gflop
        tot["floor"] += n * gflop / PEAK_TF * 1e3

        def tf(us):
            return "%.0f" % (gflop / us * 1e3) if us and us > 0 else "-"
        rows.append("| %d | %d | %d | %d | %d | %d | %dx%d | %.2f | %.0f / %s | %.0f / %s | %.0f / %s | %s / %s | %s / %s | %.1e | S %d, %d wg x %d chunks | %s |" % (
            n, cin, cout, k, s, d, xs[2], xs[3], gflop, t["mi_f"], ("%.0f" % t["my_f"]) if "my_f" in t else "-",
            t["mi_d"], ("%.0f" % t["my_d"]) if "my_d" in t else "-", t["mi_w"], "%.0f" % t["my_w"],
            tf(t["mi_d"]), tf(t.get("my_d")), tf(t["mi_w"]), tf(t["my_w"]), err, plan[4], plan[5], plan[3] // max(1, plan[4]), names[0]))
    head = ["# dense convolutions of one training step, batch [%d,3,%d,%d] (tools/conv_train_table.py)" % (N, H, W), "",
            "us per call, MIOpen / this package; TF/s likewise; f32 MFMA peak %.1f TFLOP/s." % PEAK_TF, "",
            "| x | Cin | Cout | k | s | d | plane | GFLOP | fwd us | dgrad us | wgrad us | dgrad TF/s | wgrad TF/s | wgrad rel err | wgrad plan | first layer |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    foot = ["", "totals per step (us): forward MIOpen %.0f / mine (supported) %.0f; dgrad MIOpen %.0f / mine (stride 1) %.0f; wgrad MIOpen %.0f / mine %.0f;"
            % (tot["mi_f"], tot["my_f"], tot["mi_d"], tot["my_d"], tot["mi_w"], tot["my_w"]),
            "best-of per direction: fwd %.0f, dgrad %.0f, wgrad %.0f; floor per direction %.0f us; %.1f GFLOP per direction"
            % (tot["best_f"], tot["best_d"], tot["best_w"], tot["floor"], tot["gflop"])]
    text = "\n".join(head + rows + foot) + "\n"
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
