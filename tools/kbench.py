#!/usr/bin/env python
"""Developer micro-benchmark: time individual scorer / loss kernels on resident synthetic tensors.
   python tools/kbench.py [k2] [k3] [loss] [--ids int16]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch  # noqa: E402
from mulactseg_amd import ops  # noqa: E402


def timeit(fn, n=int(os.environ.get("KBENCH_N", "30")), warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3, t[0] * 1e3


def main():
    which = [a for a in sys.argv[1:] if not a.startswith('--')] or ['k2', 'k3']
    ids = 'int64'
    if '--ids' in sys.argv:
        ids = sys.argv[sys.argv.index('--ids') + 1]
    dev = torch.device('cuda:0')
    B, C, H, W, S = 4, 20, 1024, 2048, 2048
    bufs = [make_batch(11 + i, B, C, H, W, S, ids, dev) for i in range(3)]
    if '--noids' in sys.argv:      # ablation: every id invalid -> no table lookups / LDS atomics in the scan kernels
        bufs = [(z, torch.full_like(spx, -1)) for z, spx in bufs]
    if '--oneid' in sys.argv:      # ablation: a single region -> every lane takes the 4-equal-keys path, same slot
        bufs = [(z, torch.zeros_like(spx)) for z, spx in bufs]
    invT = ops.inv_temperature(0.1)
    w = torch.linspace(0.3, 1.0, C, device=dev)
    it = [0]
    if 'k2' in which:
        out = torch.zeros((B, C), dtype=torch.int64, device=dev)
        def f():
            it[0] += 1
            ops.class_prob_sum(bufs[it[0] % 3][0], invT, out=out)
        med, mn = timeit(f)
        print("k2  median %.1f us  min %.1f us  -> %.2f TB/s" % (med, mn, B * C * H * W * 4 / med / 1e6))
    if 'k3' in which:
        ss = torch.zeros((B, S), dtype=torch.int64, device=dev)
        hh = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
        idb = {'int64': 8, 'int32': 4, 'int16': 2}[ids]
        def f():
            it[0] += 1
            z, spx = bufs[it[0] % 3]
            ops.bvsb_region_accum(z, spx, w, S, invT, score_sum=ss, hist=hh)
        med, mn = timeit(f)
        print("k3  median %.1f us  min %.1f us  -> %.2f TB/s (%s ids)" % (med, mn, B * (C * H * W * 4 + H * W * idb) / med / 1e6, ids))
    if 'fused' in which and hasattr(ops, 'single_pass_accum'):
        ps = torch.zeros((B, C), dtype=torch.int64, device=dev)
        s2 = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
        hh = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
        idb = {'int64': 8, 'int32': 4, 'int16': 2}[ids]
        def f():
            it[0] += 1
            z, spx = bufs[it[0] % 3]
            ops.single_pass_accum(z, spx, S, invT, prob_sum=ps, class_sum=s2, hist=hh)
        med, mn = timeit(f)
        print("fused median %.1f us  min %.1f us  -> %.2f TB/s (%s ids)" % (med, mn, B * (C * H * W * 4 + H * W * idb) / med / 1e6, ids))
    if 'loss' in which:
        from mulactseg_amd import synth, _lib
        N, crop, Sx = 4, 768, 2048
        frac = 0.09
        if '--frac' in sys.argv:
            frac = float(sys.argv[sys.argv.index('--frac') + 1])
        spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, Sx, frac_selected=frac) for i in range(N)])
        spx = torch.from_numpy(np.stack(spx)).to(dev)
        msk = torch.from_numpy(np.stack(msk)).to(dev)
        tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, Sx, C) for i in range(N)])).to(dev)
        bits = ops.target_bits(tgt)
        zz = 0.35 * torch.randn((N, C, crop, crop), device=dev)
        flags = _lib.LOSS_CE | _lib.LOSS_GROUP | _lib.LOSS_GROUP_ONLY_MULTI | _lib.LOSS_DECOMP
        go = torch.ones(3, device=dev)
        keep = {}
        def ffwd():
            keep['r'] = ops.partial_loss_fwd(zz, spx, msk, bits, invT, flags)
        med, mn = timeit(ffwd)
        print("loss fwd (scan+finalize+values) median %.1f us min %.1f us; selected %.3f" % (med, mn, float(msk.float().mean())))
        losses, acc, gmax = keep['r']
        def fbwd():
            ops.partial_loss_bwd(zz, spx, msk, bits, gmax, acc, go, invT, flags)
        med, mn = timeit(fbwd)
        print("loss bwd (scales+scan) median %.1f us min %.1f us -> dz write %.2f TB/s" % (med, mn, zz.numel() * 4 / med / 1e6))
    if 'augment' in which:
        import random
        from mulactseg_amd.dataloader import device_transforms as dtm
        rs = np.random.RandomState(0)
        Hh, Ww = 1024, 2048
        img = torch.from_numpy(rs.randint(0, 256, size=(Hh, Ww, 3)).astype(np.uint8)).to(dev)
        lbl = torch.from_numpy(rs.randint(0, 20, size=(Hh, Ww)).astype(np.uint8)).to(dev)
        spx = torch.from_numpy(rs.randint(0, 2048, size=(Hh, Ww)).astype(np.int16)).to(dev)
        aug = dtm.DeviceTrainAugment(rng=random.Random(0))
        t0 = time.perf_counter()
        for _ in range(50):
            aug(img, [lbl, spx])
        torch.cuda.synchronize()
        print("augment 1024x2048 -> 768x768 (picture + 2 maps): %.3f ms per sample wall (host tables + H2D + kernel)" % ((time.perf_counter() - t0) / 50 * 1e3))
        med, mn = timeit(lambda: aug(img, [lbl, spx], params=dtm.draw_params(random.Random(1), Hh, Ww, (768, 768))))
        print("augment fixed params: event median %.1f us" % med)
        try:
            from PIL import Image
            pim, plb, psp = Image.fromarray(img.cpu().numpy()), Image.fromarray(lbl.cpu().numpy()), Image.fromarray(spx.cpu().numpy().astype(np.int32)).convert('I')
            rng = random.Random(0)
            t0 = time.perf_counter()
            for _ in range(10):
                p = dtm.draw_params(rng, Hh, Ww, (768, 768))
                a = pim.resize((p['tw'], p['th']), Image.BILINEAR); b = plb.resize((p['tw'], p['th']), Image.NEAREST); c = psp.resize((p['tw'], p['th']), Image.NEAREST)
                box = (max(p['j'] - p['gap_x'], 0), max(p['i'] - p['gap_y'], 0), min(p['j'] - p['gap_x'] + 768, p['tw']), min(p['i'] - p['gap_y'] + 768, p['th']))
                t = torch.from_numpy(np.array(a.crop(box), dtype=np.uint8).transpose(2, 0, 1).copy()).float().div(255)
                np.array(b.crop(box)); np.array(c.crop(box), dtype=np.int64)
            print("Pillow on one host core (resize + crop + to-tensor, no pad/flip): %.1f ms per sample" % ((time.perf_counter() - t0) / 10 * 1e3))
        except ImportError:
            pass
    if 'k4' in which:
        n_img, Sx = 2975, 2048                                   # the whole Cityscapes pool
        g = torch.Generator(device=dev); g.manual_seed(1)
        csum = torch.randint(0, 1 << 44, (n_img, Sx, C), generator=g, device=dev, dtype=torch.int64)
        hh = torch.randint(0, 200, (n_img, Sx, C), generator=g, device=dev, dtype=torch.int32)
        w31 = torch.from_numpy(ops.weights_to_fixed31(np.linspace(0.3, 1.0, C).astype(np.float32)).view(np.int32)).to(dev)
        rank_t = torch.arange(n_img, dtype=torch.int32, device=dev)
        cost = torch.randint(1, 4, (n_img * Sx,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
        st = {}
        def fin():
            st['score'] = ops.region_finalize_weighted(csum, hh, w31, C - 1)[0]
        def keys():
            st['keys'] = ops.region_keys(st['score'], None, rank_t)
        def sort():
            st['sorted'] = ops.sort_keys_desc(st['keys'])
        def walk():
            st['sel'] = ops.budget_walk(st['sorted'], cost, rank_t, Sx, 100000, 100001)
        for name, fn in (("finalize_weighted", fin), ("region_keys", keys), ("sort_keys_desc", sort), ("budget_walk", walk)):
            med, mn = timeit(fn, n=10, warm=2)
            print("K4 %-18s median %8.1f us  (6.09 M regions)" % (name, med))
        print("selected", int(st['sel'][0].item()))


if __name__ == "__main__":
    main()
