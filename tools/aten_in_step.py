#!/usr/bin/env python
"""Which ATen operators still launch kernels inside a training step (bench batch), with shapes and device time (torch.profiler)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch.profiler import ProfilerActivity, profile
    import bench
    from mulactseg_amd.models import get_model
    from mulactseg_amd.utils.optim import FusedAdamW
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((4, 3, 768, 768), generator=g, device=dev)
    cot = None

    def step():
        nonlocal cot
        opt.zero_grad(set_to_none=True)
        out = net(x, lowres=True)
        if cot is None:
            cot = torch.randn(out.shape, generator=g, device=dev)
        (out * cot).sum().backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = getattr(e, "self_cuda_time_total", 0)
        if dt > 0 and e.key.startswith("aten::"):
            rows.append((dt / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("ATen operators with device time inside one training step [4,3,768,768] (us per step, calls per step): total %.0f us" % tot)
    for dt, n, k, sh in rows[:40]:
        print("%8.1f us %5.1f  %-28s %s" % (dt, n, k, sh))


if __name__ == "__main__":
    main()
