#!/usr/bin/env python
"""Are the gradients of a training step on TWO streams (MAS_WGRAD_STREAM=async, the default) the bits of the one-stream step, at the bench
size?  (Round 6, NOTEBOOK.md section 16.7: kernels of different streams share compute units; a kernel holding an instruction form that
is wrong beside another kernel's MFMA waves would show here.)

    python tools/stream_bits_check.py [--crop 768] [--reps 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=768)
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((4, 3, a.crop, a.crop), generator=g, device=dev)
    cot = []

    def step(mode):
        os.environ['MAS_WGRAD_STREAM'] = mode
        net.zero_grad(set_to_none=True)
        torch.manual_seed(11)                       # the dropout mask of the head
        out = net(x)
        out = out['out'] if isinstance(out, dict) else out
        if not cot:
            cot.append(torch.randn(out.shape, generator=g, device=dev))
        (out * cot[0]).sum().backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}, out.detach().clone()
    ref, oref = step('main')
    bad_total = 0
    for mode in ('main', 'async'):
        for rep in range(a.reps if mode == 'async' else 2):
            got, o = step(mode)
            bad = [n for n in ref if not torch.equal(got[n], ref[n])]
            bad_total += len(bad) + (not torch.equal(o, oref))
            print("%-5s step %d: outputs equal %s, gradient tensors differing from the first one-stream step: %d of %d %s"
                  % (mode, rep, torch.equal(o, oref), len(bad), len(ref), bad[:4]), flush=True)
    print("crop %d: %s" % (a.crop, "two-stream steps reproduce the one-stream bits" if bad_total == 0 else "MISMATCH"))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
