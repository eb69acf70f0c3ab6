#!/usr/bin/env python
"""Wall time per picture of the stage-2 pseudo-label generation LOOP (trainer/eval_save_cosplbl_prop*.py: forward at batch 1, K9 kernels,
IoU counters, one PNG per picture) on synthetic resident 1024 x 2048 pictures -- the loop, not only the kernels bench.py's stage2 leg times.

    python tools/stage2_loop_probe.py [--pictures 48]"""
import argparse
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=48)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2, 3, 4])
    a = ap.parse_args()
    from mulactseg_amd import synth
    from mulactseg_amd.models import get_model
    from mulactseg_amd.trainer import eval_save_cosplbl_prop_includeonehot as G
    dev = torch.device('cuda:0')
    C, H, W, S = 19, 1024, 2048, 2048
    torch.manual_seed(2)
    net = get_model('deeplabv3pluswn_resnet50deepstem', C + 1, 16, True, pretrained_backbone=False).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(4)
    nbuf = 4
    pics = torch.randn((nbuf, 1, 3, H, W), generator=g, device=dev)
    spx = [torch.from_numpy(synth.superpixel_map(70 + i, H, W, S)[None]).to(dev) for i in range(nbuf)]
    rs = np.random.RandomState(5)
    samples = []
    for i in range(nbuf):
        lab = torch.from_numpy(rs.randint(0, C, size=(1, H, W))).to(dev)
        trg = torch.from_numpy((rs.rand(1, S, C + 1) < 0.1).astype(np.uint8))
        trg[..., C] = 0
        sel = torch.from_numpy(rs.rand(S) < 0.3)
        trg[0, ~sel] = 0
        trg = trg.to(dev)
        msk = (trg.sum(-1) > 0)[0][spx[i][0].long()][None]
        samples.append((lab, trg, msk))
    tmp = tempfile.mkdtemp(prefix="mas_s2_")

    class Loader:
        def __init__(self, n):
            self.n, self.k = n, 0

        def __len__(self):
            return self.n

        def __next__(self):
            i = self.k % nbuf
            self.k += 1
            lab, trg, msk = samples[i]
            return {'images': pics[i], 'labels': lab, 'spx': spx[i], 'spmask': msk, 'target': trg,
                    'fnames': [["i/p%05d.png" % self.k, "l/p%05d.png" % self.k, "s/p%05d.pkl" % self.k]]}
    import hashlib
    results = {}
    for workers in a.workers:
        os.environ["MAS_STAGE2_WORKERS"] = str(workers)
        run = os.path.join(tmp, "w%d" % workers)
        os.makedirs(run)
        tr = object.__new__(G.ActiveTrainer)
        tr.args = types.SimpleNamespace(ignore_idx=255, init_checkpoint=os.path.join(run, "checkpoint01.tar"), plbl_type=None, val_batch_size=1)
        tr.net, tr.device, tr.num_classes, tr.selection_iter, tr.save_dir = net, dev, C, 1, None
        tr.inference(Loader(6))                     # warm-up (first launches of the process, the threads' streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        miou, table = tr.inference(Loader(a.pictures))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        files = sorted(f for f in os.listdir(tr._save_dir()) if f.endswith(".png"))
        h = hashlib.sha256()
        for f in files:
            h.update(f.encode())
            h.update(open(os.path.join(tr._save_dir(), f), "rb").read())
        results[workers] = (h.hexdigest(), table)
        print("MAS_STAGE2_WORKERS=%d: %d pictures in %.2f s = %.1f ms per picture; %d PNGs, sha256 over names and bytes %s"
              % (workers, a.pictures, dt, dt / a.pictures * 1e3, len(files), h.hexdigest()[:16]), flush=True)
    same = len(set(results.values())) == 1
    print("PNG files and IoU table identical for every worker count" if same else "MISMATCH between worker counts")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
