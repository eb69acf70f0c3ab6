#!/usr/bin/env python
"""Forward / input-gradient products of the training step, per layer shape: the persistent stream-K f32 kernel (csrc/conv_sk.hip; forward
with the BatchNorm statistics epilogue) against the split-bf16 kernel (csrc/conv_bx.hip; bare product) on one MI355X.
  python tools/bx_train_table.py [--shape train|train769] [--out gpurun_out/bx_train_table.md]"""
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
    ap.add_argument("--shape", default="train")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    N, H, W = {"train": (4, 768, 768), "train769": (4, 769, 769)}[args.shape]
    dev = torch.device('cuda:0')
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev)
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
    os.environ["MAS_TRAIN_CONV"] = "miopen"          # (collect the shapes through Module.__call__)
    net.train()
    with torch.no_grad():
        net(torch.randn(N, 3, H, W, device=dev))
    lines = ["# training products per layer, batch [%d,3,%d,%d]: stream-K f32 kernel vs split-bf16 kernel, us per call" % (N, H, W), "",
             "| x | Cin | Cout | k | d | H | W | GFLOP | fwd sk+stats | fwd bx | dgrad sk | dgrad bx | wgrad f32 | wgrad bx |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = collections.Counter()
    lib = _lib.load()
    for (cin, cout, k, s, d, g, xs), names in shapes.items():
        if g != 1 or s != 1 or cin < 8 or xs[2] * xs[3] < 64:
            continue
        x = torch.randn(xs, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        mult = len(names)
        with torch.no_grad():
            pk = ops.conv_sk_pack(w, 1, False)
            pd = ops.conv_sk_pack(w, 1, True)
            y, _ = ops.conv_sk(x, w, 1, d, packed=pk, stats=True)
            dy = torch.randn_like(y)
            t_f = timeit(lambda: ops.conv_sk(x, w, 1, d, packed=pk, stats=True))
            t_d = timeit(lambda: ops.conv_sk(dy, w, 1, d, dgrad=True, packed=pd))
            b_f = b_d = None
            if lib.mas_conv_bx_supported(k, 1, d, cin, cout, xs[2], xs[3]):
                bk = ops.conv_bx_pack(w, 0)
                yb = ops.conv_bx_raw(x, w, d, packed=bk)
                assert float((yb - y).abs().max()) <= 2e-5 * float(y.abs().max()), names[0]
                b_f = timeit(lambda: ops.conv_bx_raw(x, w, d, packed=bk))
            if lib.mas_conv_bx_supported(k, 1, d, cout, cin, xs[2], xs[3]):
                bd = ops.conv_bx_pack(w, 1)
                dxs = ops.conv_sk(dy, w, 1, d, dgrad=True, packed=pd)
                dxb = ops.conv_bx_raw(dy, w, d, dgrad=True, packed=bd)
                assert float((dxb - dxs).abs().max()) <= 2e-5 * float(dxs.abs().max()), names[0]
                b_d = timeit(lambda: ops.conv_bx_raw(dy, w, d, dgrad=True, packed=bd))
            os.environ["MAS_TRAIN_BX"] = "off"
            t_w = timeit(lambda: ops.conv_wgrad(x, dy, k, 1, d))
            os.environ.pop("MAS_TRAIN_BX")
            b_w = None
            if k == 1 and lib.mas_conv_wgrad_bx_supported(xs[0], cin, xs[2], xs[3], cout):
                b_w = timeit(lambda: ops.conv_wgrad_bx(x, dy))
            if k == 3 and lib.mas_conv_wgrad_bx3_supported(xs[0], cin, xs[2], xs[3], cout, d):
                wf = ops.conv_wgrad(x, dy, k, 1, d) if False else None
                os.environ["MAS_TRAIN_BX"] = "off"
                wf = ops.conv_wgrad(x, dy, k, 1, d)
                os.environ.pop("MAS_TRAIN_BX")
                wb = ops.conv_wgrad_bx3(x, dy, d)
                assert float((wb - wf).abs().max()) <= 2e-5 * float(wf.abs().max()), names[0]
                b_w = timeit(lambda: ops.conv_wgrad_bx3(x, dy, d))
        tot['w_f32'] += mult * t_w
        tot['w_best'] += mult * min(t_w, b_w if b_w else 1e9)
        flop = 2.0 * cin * k * k * y.numel()
        tot['f_sk'] += mult * t_f
        tot['d_sk'] += mult * t_d
        tot['f_best'] += mult * min(t_f, b_f if b_f else 1e9)
        tot['d_best'] += mult * min(t_d, b_d if b_d else 1e9)
        lines.append("| %d | %d | %d | %d | %d | %d | %d | %.2f | %.0f | %s | %.0f | %s | %.0f | %s |" % (
            mult, cin, cout, k, d, xs[2], xs[3], flop / 1e9, t_f, "%.0f" % b_f if b_f else "-", t_d, "%.0f" % b_d if b_d else "-",
            t_w, "%.0f" % b_w if b_w else "-"))
    lines += ["", "per step (us, stride-1 layers): forward stream-K %.0f, best of the two per layer %.0f; input gradient stream-K %.0f, best %.0f; weight gradient f32 kernel %.0f, best %.0f"
              % (tot['f_sk'], tot['f_best'], tot['d_sk'], tot['d_best'], tot['w_f32'], tot['w_best'])]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
