#!/usr/bin/env python
"""Time the stage-1 training step (model fwd + fused losses + bwd + AdamW, batch [4,3,crop,crop]) under the training-convolution
modes of ops.conv_train_plan:  python tools/train_step_probe.py [--crop 768] [--modes miopen,auto,own] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=768)
    ap.add_argument("--modes", default="miopen,auto,own")
    ap.add_argument("--streams", default="side")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--adamw", default="own", choices=["own", "aten"])
    ap.add_argument("--prio", type=int, default=0, help="run the step on a stream of this priority (-1 = high) instead of the default stream")
    args = ap.parse_args()
    from mulactseg_amd import synth
    from mulactseg_amd.models import deeplab, get_model
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    dev = torch.device('cuda:0')
    N, C, S, crop = 4, 20, 2048, args.crop
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx = torch.from_numpy(np.stack(spx)).to(dev)
    msk = torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=True)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    images = torch.randn((N, 3, crop, crop), generator=g, device=dev)
    out = {}
    for mode in args.modes.split(","):
        for st in args.streams.split(","):
            os.environ["MAS_TRAIN_CONV"] = mode
            os.environ["MAS_WGRAD_STREAM"] = st
            torch.manual_seed(0)
            net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).train()
            groups = [{'params': list(net.backbone.parameters()), 'lr': 2e-5}, {'params': list(net.classifier.parameters()), 'lr': 2e-4}]
            if args.adamw == "own":
                from mulactseg_amd.utils.optim import FusedAdamW
                opt = FusedAdamW(groups, lr=2e-5, weight_decay=1e-5)
            else:
                opt = torch.optim.AdamW(groups, lr=2e-5, weight_decay=1e-5, fused=True)

            def step():
                opt.zero_grad(set_to_none=True)
                total, _, _, _ = crit.weighted_lowres(net(images, lowres=True), (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)
                total.backward()
                opt.step()
                return total
            ctx = torch.cuda.stream(torch.cuda.Stream(dev, priority=args.prio)) if args.prio != 0 else None
            if ctx is not None:
                torch.cuda.synchronize()
                ctx.__enter__()
            for _ in range(3):
                loss = step()
            deeplab.path_report(reset=True)
            step()
            paths = deeplab.path_report(reset=True)
            torch.cuda.synchronize()
            if os.environ.get("MAS_PROBE_ALLOC"):       # per-step allocator counters (diagnostic)
                for k in range(12):
                    a = torch.cuda.memory_stats(dev)
                    step()
                    torch.cuda.synchronize()
                    b = torch.cuda.memory_stats(dev)
                    print("   step %2d: device mallocs %d frees %d retries %d reserved %.2f GB active %.2f GB inactive_split %.2f GB" % (
                        k, b["num_device_alloc"] - a["num_device_alloc"], b["num_device_free"] - a["num_device_free"],
                        b["num_alloc_retries"] - a["num_alloc_retries"], b["reserved_bytes.all.current"] / 2 ** 30,
                        b["active_bytes.all.current"] / 2 ** 30, b["inactive_split_bytes.all.current"] / 2 ** 30), flush=True)
            seg0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)      # hipMalloc calls of the caching allocator so far
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # the host has queued all steps (nothing in a step reads the device)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            out["%s/%s" % (mode, st)] = {"ms_per_step": ms, "host_ms_per_step": host_ms, "loss": float(loss), "conv_paths": paths.get("conv_bn_act")}
            segs = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0) - seg0
            print(mode, st, "%.2f ms" % ms, "(host %.2f ms)" % host_ms, float(loss), "hipMallocs in the timed steps: %d, reserved %.1f GB" %
                  (segs, torch.cuda.memory_reserved(dev) / 2 ** 30), paths.get("conv_bn_act"), flush=True)
            if ctx is not None:
                ctx.__exit__(None, None, None)
            del net, opt
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
