#!/usr/bin/env python
"""The stage-1 loss legs alone, batch [4,20,crop,crop] (BASELINE configs[1]): forward + backward of
  full   -- FusedPartialLabelLoss on full-resolution logits (bench.py train_iter.loss_only),
  lowres -- FusedPartialLabelLoss.weighted_lowres on the model's quarter-resolution logits (what the trainer runs),
each through the fused entry points (one library call per direction) and, with --stepwise, through the step-by-step entry points of
rounds 1-4 (target_bits, scan, group_finalize, loss_values / loss_scales, scan, fix_to_float) for the A/B.
  python tools/loss_probe.py [--crop 768] [--iters 50] [--stepwise] [--forms full,lowres]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd import _lib, ops, synth                       # noqa: E402
from mulactseg_amd.utils.loss import FusedPartialLabelLoss       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=768)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--stepwise", action="store_true")
    ap.add_argument("--forms", default="full,lowres")
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    N, C, S, crop = 4, 20, 2048, args.crop
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx = torch.from_numpy(np.stack(spx)).to(dev)
    msk = torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    q = ((crop - 1) // 2) // 2 + 1
    z = (0.35 * torch.randn((N, C, crop, crop), generator=g, device=dev)).requires_grad_(True)
    zq = (0.35 * torch.randn((N, C, q, q), generator=g, device=dev)).requires_grad_(True)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=False)
    invT = ops.inv_temperature(0.1)
    flags = crit.flags
    wts = torch.tensor([16.0, 8.0, 1.0], device=dev)
    go3 = torch.tensor([16.0, 8.0, 1.0], device=dev)
    go1 = torch.ones(1, device=dev)

    def full_fused():
        z.grad = None
        gr, ce, mc = crit(z, tgt, spx, msk)
        (16.0 * ce + 8.0 * mc + gr).backward()

    def low_fused():
        zq.grad = None
        crit.weighted_lowres(zq, (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)[0].backward()

    def full_step():
        bits = ops.target_bits(tgt)
        _, acc, gmax = ops.partial_loss_fwd(z.detach(), spx, msk, bits, invT, flags)
        ops.partial_loss_bwd(z.detach(), spx, msk, bits, gmax, acc, go3, invT, flags)

    def low_step():
        bits = ops.target_bits(tgt)
        _, acc, gmax = ops.partial_loss_fwd_lowres(zq.detach(), (crop, crop), spx, msk, bits, invT, flags, weights=wts)
        ops.partial_loss_bwd_lowres(zq.detach(), (crop, crop), spx, msk, bits, gmax, acc, go1, invT, flags, weights=wts)

    def full_raw():          # the fused entry points without the autograd machinery around them
        _, st = ops.partial_loss_fwd_fused(z.detach(), None, spx, msk, invT, flags, targets=tgt)
        ops.partial_loss_bwd_fused(z.detach(), None, spx, msk, st, go3, invT)

    def low_raw():
        _, st = ops.partial_loss_fwd_fused(zq.detach(), (crop, crop), spx, msk, invT, flags, targets=tgt, weights=wts)
        ops.partial_loss_bwd_fused(zq.detach(), (crop, crop), spx, msk, st, go1, invT, weights=wts)

    runs = []
    if "full" in args.forms:
        runs += [("full-resolution logits, loss modules (autograd)", full_fused), ("full-resolution logits, fused entry points, no autograd", full_raw)]
        if args.stepwise:
            runs.append(("full-resolution logits, step-by-step entry points, no autograd", full_step))
    if "lowres" in args.forms:
        runs += [("quarter-resolution logits, weighted_lowres (autograd)", low_fused), ("quarter-resolution logits, fused entry points, no autograd", low_raw)]
        if args.stepwise:
            runs.append(("quarter-resolution logits, step-by-step entry points, no autograd", low_step))
    for name, fn in runs:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print("%-72s %.3f ms per fwd+bwd (host issue %.3f ms)" % (name, t / args.iters * 1e3, t_host / args.iters * 1e3))


if __name__ == "__main__":
    main()
