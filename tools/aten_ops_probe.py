#!/usr/bin/env python
"""Which ATen ops (not this package's kernels) a training step still launches: torch.profiler over one steady-state step of
tools/train_step_probe.py's loop, grouped by op name and input shapes.   python tools/aten_ops_probe.py [--crop 768]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=768)
    args = ap.parse_args()
    from mulactseg_amd import synth
    from mulactseg_amd.models import get_model
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    dev = torch.device('cuda:0')
    N, C, S, crop = 4, 20, 2048, args.crop
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx = torch.from_numpy(np.stack(spx)).to(dev)
    msk = torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=True)
    images = torch.randn((N, 3, crop, crop), device=dev)
    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).train()
    opt = torch.optim.AdamW([{'params': net.backbone.parameters(), 'lr': 2e-5}, {'params': net.classifier.parameters(), 'lr': 2e-4}],
                            lr=2e-5, weight_decay=1e-5, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        total, _, _, _ = crit.weighted_lowres(net(images, lowres=True), (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)
        total.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0.0)
        self_dt = getattr(e, "self_device_time_total", None)
        if self_dt is None:
            self_dt = getattr(e, "self_cuda_time_total", 0.0)
        if e.key.startswith("aten::") and self_dt > 0:
            rows.append((self_dt, e.count, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print("self device us | calls | op | input shapes")
    for r in rows[:60]:
        print("%9.1f | %3d | %s | %s" % r)
    print("total self device time of aten ops: %.1f us" % sum(r[0] for r in rows))


if __name__ == "__main__":
    main()
