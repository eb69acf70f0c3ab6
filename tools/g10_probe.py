#!/usr/bin/env python
"""Where the GPU training step deviates from the executed reference (tests/golden/g10_train.npz) and from itself:
  python tools/g10_probe.py det     two identical training forwards, module by module: the first module whose output differs
  python tools/g10_probe.py grads   relative error of every parameter gradient (own kernels / MIOpen convolutions) against the fixture"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_train_golden as t  # noqa: E402


def det():
    from mulactseg_amd.models import get_model
    dev = torch.device('cuda:0')
    torch.manual_seed(13)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = torch.randn(N, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(4), device=dev)
    runs = []
    for rep in range(3):
        outs = {}
        hooks = [m.register_forward_hook(lambda m, i, o, n=n: outs.__setitem__(n, o.detach().clone()) if torch.is_tensor(o) else None)
                 for n, m in net.named_modules() if n]
        z = net(x, lowres=True)
        z.square().mean().backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        for p in net.parameters():
            p.grad = None
        for h in hooks:
            h.remove()
        runs.append((outs, grads))
    for a, b, tag in ((1, 2, "run 2 vs run 3"), (0, 1, "run 1 vs run 2")):
        print(tag)
        shown = 0
        for n in runs[a][0]:
            d = float((runs[a][0][n] - runs[b][0][n]).abs().max())
            if d > 0:
                print("   fwd", n, d, float(runs[a][0][n].abs().max()))
                shown += 1
                if shown > 8:
                    break
        bad = [(float((runs[a][1][n] - runs[b][1][n]).abs().max()), n) for n in runs[a][1]]
        bad = [v for v in bad if v[0] > 0]
        print("   gradients that differ: %d of %d" % (len(bad), len(runs[a][1])), bad[:5])


def grads():
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    g = np.load(t.GOLDEN)
    x, tgt, spx, msk = t._inputs(g)
    dev = torch.device('cuda:0')
    res = {}
    for mode in ("own", "miopen"):
        os.environ["MAS_TRAIN_CONV"] = mode
        net, opt, sched = t._build(g, dev)
        xt, tt, ts, tm = (torch.from_numpy(a).to(dev) for a in (x, tgt, spx, msk))
        crit = FusedPartialLabelLoss(int(g['S']), float(g['temp']), float(g['temp']), sync_normalisers=False)
        zq = net(xt, lowres=True)
        total, group, ce, mc = crit.weighted_lowres(zq, (int(g['H']), int(g['W'])), tt, ts, tm, 16.0, 8.0, 1.0)
        total.backward()
        print(mode, "logits", float(np.abs(zq.detach().cpu().numpy() - g['quarter1']).max()), "loss", float(total), g['losses1'][0])
        rows = []
        num = den = 0.0
        for i, (n, p) in enumerate(net.named_parameters()):
            ref = g['grad_%03d' % i].astype(np.float64)
            got = t.sub256(p.grad.detach().cpu().numpy()).astype(np.float64)
            num += float(((got - ref) ** 2).sum())
            den += float((ref ** 2).sum())
            rows.append((float(np.sqrt(((got - ref) ** 2).sum() / max(1e-60, (ref ** 2).sum()))), n))
        res[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        print(mode, "rel L2 of all cuts %.3e" % (num / den) ** 0.5)
        for r in rows[::12]:
            print("   %.2e  %s" % r)
    num = sum(float(((res["own"][n].double() - res["miopen"][n].double()) ** 2).sum()) for n in res["own"])
    den = sum(float((res["miopen"][n].double() ** 2).sum()) for n in res["own"])
    print("own vs miopen, all elements: rel L2 %.3e" % (num / den) ** 0.5)


if __name__ == "__main__":
    {"det": det, "grads": grads}[sys.argv[1]]()
