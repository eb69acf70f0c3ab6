#!/usr/bin/env python
"""The scores of an acquisition round with the model in the loop, pool batches on 1 / 2 / 3 streams, at the bench picture size: the same bits?
(tests/test_selectors_gpu.py holds the small case; this is bench.py's own synthetic pool and model, --images pictures of 1024 x 2048.)

    python tools/pool_streams_bits.py [--images 400]"""
import argparse
import hashlib
import os
import sys
import tempfile
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=400)
    a = ap.parse_args()
    import bench
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd.models import get_model
    from mulactseg_amd.synth_pool import SyntheticPool
    dev = torch.device('cuda:0')
    B, C, H, W, S = 4, 20, 1024, 2048, 2048
    torch.manual_seed(1)
    net = bench.ModelOnRotatingPictures(get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).eval(), B, H, W, dev)
    pool = SyntheticPool(a.images, H, W, S, dev, shard=(0, a.images))
    out = {}
    for n in ("1", "2", "2", "3", "1"):
        os.environ["MAS_POOL_STREAMS"] = n
        tmp = tempfile.mkdtemp(prefix="mas_psb_")
        args = types.SimpleNamespace(val_batch_size=B, val_num_workers=0, nseg=S, active_method='pixbal', num_classes=C - 1, ce_temp=0.1,
                                     cls_weight_coeff=6.0, method='active_joint_multi_predignore_lossdecomp', save_scores=False,
                                     fair_counting=True, or_labeling=True, model_save_dir=tmp, finetune_itrs=1,
                                     wandb=types.SimpleNamespace(log=lambda *x, **k: None))
        sel = banignore.RegionSelector(args)
        tr = types.SimpleNamespace(net=net, device=dev, model_save_dir=tmp, selection_iter=1)
        net.k = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scores, hist = sel.calculate_scores_tensor(tr, pool, want_hist=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h = hashlib.sha256()
        for t in (scores, hist, sel.cls_weight):
            h.update(t.detach().cpu().numpy().tobytes())
        print("MAS_POOL_STREAMS=%s: %d pictures in %.2f s (%.2f ms per batch of 4), sha256 of scores + histograms + class weights %s"
              % (n, a.images, dt, dt / (a.images / B) * 1e3, h.hexdigest()[:16]), flush=True)
        out.setdefault(n, set()).add(h.hexdigest())
    same = len(set().union(*out.values())) == 1
    print("all runs give the same bits" if same else "MISMATCH between runs")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
