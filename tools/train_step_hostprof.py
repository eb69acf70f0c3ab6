#!/usr/bin/env python
"""Where the HOST time of a training step goes (cProfile over 10 steps of tools/train_step_probe.py's step, own kernels, no device
reads): the step is ~19 ms of Python / ctypes / autograd per ~23.5 ms of GPU time, so the host is the next bound.
    python tools/train_step_hostprof.py [--crop 768] [--top 45]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=768)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from mulactseg_amd import synth
    from mulactseg_amd.models import get_model
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    from mulactseg_amd.utils.optim import FusedAdamW
    dev = torch.device('cuda:0')
    N, C, S, crop = 4, 20, 2048, args.crop
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx, msk = torch.from_numpy(np.stack(spx)).to(dev), torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=True)
    images = torch.randn((N, 3, crop, crop), generator=torch.Generator(device=dev).manual_seed(5), device=dev)
    torch.manual_seed(0)
    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).train()
    opt = FusedAdamW([{'params': list(net.backbone.parameters()), 'lr': 2e-5}, {'params': list(net.classifier.parameters()), 'lr': 2e-4}],
                     lr=2e-5, weight_decay=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        total, _, _, _ = crit.weighted_lowres(net(images, lowres=True), (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)
        total.backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    # forward / backward / optimizer host times with an idle GPU queue in front (the host never waits for the device)
    parts = {'forward': 0.0, 'backward': 0.0, 'optimizer': 0.0}
    for _ in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        total, _, _, _ = crit.weighted_lowres(net(images, lowres=True), (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)
        t1 = time.perf_counter()
        total.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        parts['forward'] += t1 - t0; parts['backward'] += t2 - t1; parts['optimizer'] += t3 - t2
    print("host ms per step:", {k: round(v / args.steps * 1e3, 2) for k, v in parts.items()}, flush=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(args.top)
    print(s.getvalue())


if __name__ == "__main__":
    main()
