#!/usr/bin/env python
"""G11 forward (train mode, step 1), module by module, against the same forward in float64 on the host: relative L2 error of every
block output for (a) the reference arithmetic (this package's modules as plain PyTorch f32 ops on the host -- bit-equal to the
fixture at one thread), (b) the own kernels on the GPU, (c) MIOpen convolutions on the GPU.  Shows WHERE an implementation loses
accuracy against exact arithmetic.    python tools/g11_layers.py [--fixture g10]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_train_golden as t  # noqa: E402


def pick(n):
    return n in ('backbone.conv1', 'backbone.bn1', 'backbone.maxpool', 'classifier.aspp', 'classifier.project', 'classifier') or \
        (n.startswith('backbone.layer') and n.count('.') == 2) or (n.startswith('classifier.aspp.convs.') and n.count('.') == 3)


def run(g, device, dtype, mode=None):
    if mode is not None:
        os.environ["MAS_TRAIN_CONV"] = mode
    try:
        net, _, _ = t._build(g, device)
        net = net.to(dtype)
        x, _, _, _ = t._inputs(g)
        outs = {}
        hooks = [m.register_forward_hook(lambda m, i, o, n=n: outs.__setitem__(n, o.detach().double().cpu().numpy()) if torch.is_tensor(o) else None)
                 for n, m in net.named_modules() if n and pick(n)]
        net(torch.from_numpy(x).to(device=device, dtype=dtype), lowres=True)       # (with autograd: the training kernels, ops._ConvTrain)
        for h in hooks:
            h.remove()
        return outs
    finally:
        os.environ.pop("MAS_TRAIN_CONV", None)


def main():
    g = np.load(t.GOLDEN if '--fixture' in sys.argv and sys.argv[sys.argv.index('--fixture') + 1] == 'g10' else t.GOLDEN11)
    exact = run(g, 'cpu', torch.float64)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    rows = {'reference f32 (host, 1 thread)': run(g, 'cpu', torch.float32)}
    torch.set_num_threads(threads)
    if torch.cuda.is_available():
        rows['own kernels'] = run(g, 'cuda:0', torch.float32, 'own')
        rows['MIOpen convolutions'] = run(g, 'cuda:0', torch.float32, 'miopen')
    names = list(rows)
    print("| module | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for n in exact:
        if any(n not in rows[k] for k in names):        # (a module the fused GPU path does not call as a module)
            continue
        den = float(np.sqrt((exact[n] ** 2).sum()))
        cells = []
        for k in names:
            d = rows[k][n] - exact[n]
            cells.append("%.2e (max %.1e)" % (float(np.sqrt((d ** 2).sum())) / den, float(np.abs(d).max())))
        print("| %s | %s |" % (n, " | ".join(cells)))


if __name__ == "__main__":
    main()
