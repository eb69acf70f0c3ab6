#!/usr/bin/env python
"""Work-splitting plan of mas_conv_bx_train per layer shape of the training step: forward and input-gradient product at ksplit 1..8
(and both 3x3 tile shapes) against the stream-K f32 kernel, with the plan the library picks.
  python tools/bx_splitk_sweep.py [--shape train|train769] [--out gpurun_out/bx_splitk.md]"""
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
    ap.add_argument("--all", action="store_true", help="also the layers whose launches fill the chip (>= 1024 workgroups unsplit)")
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
    os.environ["MAS_TRAIN_CONV"] = "miopen"
    net.train()
    with torch.no_grad():
        net(torch.randn(N, 3, H, W, device=dev))
    lib = _lib.load()
    lines = ["# mas_conv_bx_train: us per call by ksplit (3x3: tile 8x32 / 16x16 / flat where it fits), batch [%d,3,%d,%d]; sk = the stream-K f32 kernel; * = the library's plan" % (N, H, W), "",
             "| x | role | K | M | k | d | H | W | wg@1 | sk | " + " | ".join("ks%d" % i for i in range(1, 9)) + " | plan | best |",
             "|---|---|---|---|---|---|---|---|---|---|" + "---|" * 10]
    tot = collections.Counter()
    for (cin, cout, k, s, d, g, xs), names in shapes.items():
        if g != 1 or s != 1 or cin < 8 or xs[2] * xs[3] < 64:
            continue
        mult = len(names)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        for role in (0, 1):
            K, M = (cout, cin) if role else (cin, cout)
            if not lib.mas_conv_bx_supported(k, 1, d, K, M, xs[2], xs[3]):
                continue
            a = torch.randn((xs[0], K, xs[2], xs[3]), device=dev)
            ks_plan, tw_plan, _ = ops.conv_bx_train_plan(a.shape, w.shape, d, bool(role))
            _, _, wg1 = ops.conv_bx_train_plan(a.shape, w.shape, d, bool(role))
            pk = ops.conv_bx_pack(w, role)
            with torch.no_grad():
                psk = ops.conv_sk_pack(w, 1, bool(role))
                t_sk = timeit(lambda: ops.conv_sk(a, w, 1, d, dgrad=bool(role), packed=psk, stats=not role) if not role else ops.conv_sk(a, w, 1, d, dgrad=True, packed=psk))
                ref = ops.conv_bx_raw(a, w, d, dgrad=bool(role), packed=pk, ksplit=1, tile_w=32)
                nch = -(-K // (32 if k == 1 else 8))
                unsplit_wgs = wg1 // max(ks_plan, 1)
                if not args.all and unsplit_wgs >= 1024 and k == 1:
                    continue
                cells, best, t_plan = [], (1e9, None), None
                for ks in range(1, 9):
                    if ks > 1 and nch // ks < 2:
                        cells.append("-")
                        continue
                    ts = []
                    flat_ok = k == 3 and xs[3] >= 8 and (((256 + xs[3] - 2) // xs[3] + 1) + 2 * d) * (xs[3] + 2 * d) <= 608
                    for tw in (((32, 16, 1) if flat_ok else (32, 16)) if k == 3 else (32,)):
                        y = ops.conv_bx_raw(a, w, d, dgrad=bool(role), packed=pk, ksplit=ks, tile_w=tw)
                        assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (names[0], ks, tw)
                        t = timeit(lambda: ops.conv_bx_raw(a, w, d, dgrad=bool(role), packed=pk, ksplit=ks, tile_w=tw))
                        ts.append(t)
                        if t < best[0]:
                            best = (t, (ks, tw))
                        if ks == ks_plan and tw == tw_plan:
                            t_plan = t
                    cells.append("/".join("%.0f" % t for t in ts) + ("*" if ks == ks_plan else ""))
            tot['sk'] += mult * t_sk
            tot['ks1'] += mult * timeit(lambda: ops.conv_bx_raw(a, w, d, dgrad=bool(role), packed=pk, ksplit=1, tile_w=32))
            tot['plan'] += mult * t_plan
            tot['best'] += mult * best[0]
            lines.append("| %d | %s | %d | %d | %d | %d | %d | %d | %d | %.0f | %s | %.0f (ks %d, tw %d) | %.0f (ks %d, tw %d) |" % (
                mult, "dgrad" if role else "fwd", K, M, k, d, xs[2], xs[3], unsplit_wgs, t_sk, " | ".join(cells), t_plan, ks_plan, tw_plan,
                best[0], best[1][0], best[1][1]))
    lines += ["", "listed layers, per step (us): stream-K f32 %.0f; split-bf16 unsplit 8x32 %.0f; library plan %.0f; best per layer %.0f"
              % (tot['sk'], tot['ks1'], tot['plan'], tot['best'])]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
