#!/usr/bin/env python
"""bench.py -- MulActSeg hot path on MI355X: superpixels scored/sec (+ train-iter images/sec).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`,
  one rank per GPU; the pool of unlabeled images is sharded across ranks (no data-path collective in
  the scan; the two tiny exchanges of the acquisition round -- class sums, scores -- are RCCL
  collectives inside the timed region).

Workload "acquisition-scan" (BASELINE.json metric "superpixels scored/sec"): one step = one reference
batch (val_batch_size = 4 pool images, logits [4,20,1024,2048] f32 + superpixel ids, already resident
in HBM) through the scorer hot path of the PixBal + ban-ignore selector
(reference: active_selection/my_bvsb_predclsbal_pwr_banignore.py:35-84): class-prior sums, per-superpixel
margin sums and arg-max-class histograms in ONE scan of the logits (k_single_pass, the product default);
after the K timed steps the weighted means + ban of all scored regions (k_region_finalize_weighted) and their
ordering + budgeted selection walk (K4, budget scaled to the scored share of the 2975-image pool) run once,
inside the timed region.  value = steps * 4 * 2048 * n_gpus / seconds.
The reference-structured two-pass kernels (K2, K1+K3) are timed separately and reported under "two_pass".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable


EVENT_EVERY = 4       # HIP-event pairs around every 4th scan launch of the timed region


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4, help="pool images per step (reference val_batch_size)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=20, help="logit channels (19 classes + undefined)")
    ap.add_argument("--nseg", type=int, default=2048)
    ap.add_argument("--id-dtype", default="int64", choices=["int64", "int32", "int16"],
                    help="superpixel id element type (the reference data layer yields int64)")
    ap.add_argument("--ramp", type=int, default=100, help="untimed launches before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--nbuf", type=int, default=3, help="distinct resident batches rotated through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary train-iter measurement")
    ap.add_argument("--train-steps", type=int, default=8)
    ap.add_argument("--acq-steps", type=int, default=8, help="steps of the secondary model-forward + scan measurement")
    ap.add_argument("--crop", type=int, default=768, help="training crop (reference: 768, transform.py:107)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline budget")
    return ap.parse_args()


def make_batch(seed, B, C, H, W, S, id_dtype, device):
    """Synthetic Cityscapes-shaped batch generated on the device (plumbing): cosine-like logits with a
    blocky class layout, jittered-grid superpixel map (mulactseg_amd.synth)."""
    from mulactseg_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    z = 0.35 * torch.randn((B, C, H, W), generator=g, device=device, dtype=torch.float32)
    cm = torch.randint(0, C, (B, 1, H // 32 + 1, W // 32 + 1), generator=g, device=device)
    cm = cm.repeat_interleave(32, 2).repeat_interleave(32, 3)[:, :, :H, :W]
    z.scatter_add_(1, cm, torch.full_like(cm, 0.6, dtype=torch.float32))
    spx = np.stack([synth.superpixel_map(seed * 131 + i, H, W, S) for i in range(B)])
    return z.contiguous(), torch.from_numpy(spx).to(getattr(torch, id_dtype)).to(device)


def cpu_baseline(args, budget_s):
    """Reference CPU path ("port": oracle/port.py, the torch-CPU restatement pinned bit-exact to the
    executed reference) on a bounded sample of the same workload, all host cores."""
    from mulactseg_amd import synth
    from oracle import port
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    C, H, W, S = args.classes, args.height, args.width, args.nseg

    def sample(n):
        z = torch.from_numpy(synth.logits(7, n, C, H, W))
        spx = torch.from_numpy(np.stack([synth.superpixel_map(900 + i, H, W, S) for i in range(n)]))
        return z, spx

    def run(z, spx):
        """calculate_scores (both passes + ban) -> tuple list -> sorted(reverse=True) -> budget walk, as the reference does"""
        n = z.shape[0]
        r = port.pixbal_scores(z, spx, args.batch, 0.1, 6.0, S, ban_ignore=True)
        im_idx = [["img_%05d.png" % i, "lbl_%05d.png" % i, "spx_%05d.pkl" % i] for i in range(n)]
        suppix = {k[2]: list(range(S)) for k in im_idx}
        tuples = port.score_list(im_idx, suppix, r['scores'])
        return port.select_regions(tuples, max(1, int(100000 * n / 2975)))

    z, spx = sample(1)
    t0 = time.perf_counter()
    run(z, spx)
    t1 = time.perf_counter() - t0
    n = int(max(1, min(16, budget_s // max(t1, 1e-3))))
    if n > 1:
        z, spx = sample(n)
        t0 = time.perf_counter()
        run(z, spx)
        t1 = time.perf_counter() - t0
    return {"value": n * S / t1, "unit": "superpixels/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic %dx%dx%d images, nseg %d: both passes + ban + tuple list + sort + budget walk "
                      "(oracle/port.py, torch %s CPU, %d threads), %.1f s" % (n, C, H, W, S, torch.__version__, cores, t1)}


def train_iter_bench(args, dev, world):
    """Secondary metric "train-iter images/sec" (BASELINE.json configs[1]): stage-1 step on a
    [4,20,crop,crop] batch.  (a) loss-only: fused partial-label losses fwd+bwd on resident logits;
    (b) full iteration: DeepLabv3+WN/ResNet50-deepstem fwd + losses + bwd + AdamW (fp32, random init)."""
    from mulactseg_amd import synth
    from mulactseg_amd.models import get_model
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    N, C, S, crop = 4, args.classes, args.nseg, args.crop
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx = torch.from_numpy(np.stack(spx)).to(dev)
    msk = torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=world > 1 or torch.distributed.is_initialized())     # global 1 + n over the data-parallel batch
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    z = (0.35 * torch.randn((N, C, crop, crop), generator=g, device=dev)).requires_grad_(True)

    def loss_step(logits):
        group, ce, mc = crit(logits, tgt, spx, msk)
        return 16.0 * ce + 8.0 * mc + 1.0 * group

    for _ in range(3):
        loss_step(z).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        z.grad = None
        loss_step(z).backward()
    torch.cuda.synchronize()
    loss_ms = (time.perf_counter() - t0) / 20 * 1e3
    loss_bytes = N * C * crop * crop * 4 * 3 + N * crop * crop * 9 * 2      # fwd read z, bwd read z + write dz, ids+mask twice

    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).train()
    opt = torch.optim.AdamW([{'params': net.backbone.parameters(), 'lr': 2e-5},
                             {'params': net.classifier.parameters(), 'lr': 2e-4}], lr=2e-5, weight_decay=1e-5, fused=True)
    if world > 1 or torch.distributed.is_initialized():       # data parallel as the trainers run it: gradients all-reduced over RCCL
        net = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index], output_device=dev.index)
    images = torch.randn((N, 3, crop, crop), generator=g, device=dev)

    def full_step():
        opt.zero_grad(set_to_none=True)
        (loss_step(net(images)) * world).backward()
        opt.step()

    for _ in range(2):
        full_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        full_step()
    torch.cuda.synchronize()
    it_ms = (time.perf_counter() - t0) / args.train_steps * 1e3
    return {"metric": "train-iter images/sec", "value": N * world / (it_ms * 1e-3), "unit": "images/s", "ms_per_iter": it_ms,
            "config": {"workload": "stage-1 step: DeepLabv3+WN/ResNet50-deepstem fwd+bwd (MIOpen fp32 + HIP memory-bound layers) + fused "
                                   "partial-label losses (HIP) + AdamW" + ("; DistributedDataParallel over RCCL, global loss normalisers" if world > 1 else ""), "batch": [N, 3, crop, crop], "logits": [N, C, crop, crop], "nseg": S,
                       "selected_fraction": float(msk.float().mean())},
            "loss_only": {"ms_fwd_bwd": loss_ms, "algorithmic_GBs": loss_bytes / (loss_ms * 1e-3) / 1e9,
                          "bytes": loss_bytes, "note": "includes ~8 small launches and the autograd glue"}}


def acquisition_with_model_bench(args, dev, world):
    """the acquisition metric "also with the model forwards".  One step = eval-mode forward of
    DeepLabv3+WN/ResNet50-deepstem (fp32, random init) on a [B,3,H,W] pool batch + the single-pass scan of its logits.
    The reference structure needs two forwards per image (class prior, then scores); the single-pass scan needs one."""
    from mulactseg_amd import ops
    from mulactseg_amd.models import get_model
    B, C, H, W, S = args.batch, args.classes, args.height, args.width, args.nseg
    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    images = torch.randn((B, 3, H, W), generator=g, device=dev)
    _, spx = make_batch(4242, B, C, H, W, S, args.id_dtype, dev)
    invT = ops.inv_temperature(0.1)
    prob = torch.zeros((B, C), dtype=torch.int64, device=dev)
    csum = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
    hist = torch.zeros((B, S, C), dtype=torch.int32, device=dev)

    def step():
        with torch.no_grad():
            z = net(images)
        ops.single_pass_accum(z.contiguous(), spx, S, invT, prob_sum=prob, class_sum=csum, hist=hist)

    for _ in range(4):              # MIOpen's find runs on the first calls of every new shape
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.acq_steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.acq_steps * 1e3
    return {"metric": "superpixels scored/sec incl. model forward", "value": B * S * world / (ms * 1e-3), "unit": "superpixels/s",
            "ms_per_batch": ms, "forwards_per_image": 1,
            "config": {"workload": "eval forward (MIOpen fp32) of [%d,3,%d,%d] + single-pass scan; the reference structure runs the "
                                   "forward twice per pool image" % (B, H, W)}}


def stage2_bench(args, dev):
    """BASELINE.json config 4: stage-2 pseudo-label generation for one 1024x2048 image -- (a) whole-image forward
    (quarter-resolution features, interpolated inside the K9 kernels), (b) the sliding-window ensemble (crop 800,
    stride 2/3: 8 windows, features summed at full resolution on the device).  15 % of the regions are selected."""
    import torch.nn.functional as F
    from mulactseg_amd import ops, synth
    from mulactseg_amd.models import get_model
    from mulactseg_amd.utils.sliding_evaluator_plbl import SlidingEval
    C, H, W, S = args.classes, args.height, args.width, args.nseg
    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    image = torch.randn((1, 3, H, W), generator=g, device=dev)
    spx = torch.from_numpy(synth.superpixel_map(77, H, W, S))[None].to(dev)
    tgt = torch.from_numpy(synth.multi_hot_targets(78, S, C))[None].to(dev)
    chosen = torch.from_numpy(np.random.RandomState(79).choice(S, size=int(0.15 * S), replace=False)).to(dev)
    lut = torch.zeros(S, dtype=torch.bool, device=dev)
    lut[chosen] = True
    msk = lut[spx]
    evaluator = SlidingEval(net, 800, 2 / 3, class_number=C)

    def whole():
        with torch.no_grad():
            feats, out = net.feat_forward_lowres(image)
            return ops.stage2_pseudo_labels(feats.contiguous(), out.contiguous(), tgt, msk, spx, True)

    def sliding():
        with torch.no_grad():
            feats, out = evaluator(image)
            return ops.stage2_pseudo_labels(F.normalize(feats[None], dim=1, p=2), out[None].contiguous(), tgt, msk, spx, True)

    res = {}
    for name, fn in (("whole_image", whole), ("sliding_800", sliding)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            lab = fn()
        torch.cuda.synchronize()
        res[name] = {"ms_per_image": (time.perf_counter() - t0) / 3 * 1e3, "labelled_fraction": float((lab != 255).float().mean())}
    return {"metric": "stage-2 pseudo-label generation, ms per 1024x2048 image (model forward + K9 kernels)", **res}


def pmc_traffic(kernel, default_shape):
    """HBM bytes per launch of `kernel` from the latest committed rocprofv3 --pmc summary
    (profiles/r*/..pmc_traffic.json, produced by profiles/summarize.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this same command, gfx950 FETCH_SIZE x2 correction applied).  None when the bench shape is not the
    profiled default shape or no summary is present: PMC counters cannot be read from inside this process."""
    if not default_shape:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_traffic.json")))
    for f in reversed(files):
        try:
            k = json.load(open(f))["kernels"].get(kernel)
            if k:
                return k["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_ddp = os.environ.get("MAS_BENCH_FORCE_DDP") == "1" and "RANK" in os.environ      # developer check of the N > 1 code path
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from mulactseg_amd import ops
    B, C, H, W, S = args.batch, args.classes, args.height, args.width, args.nseg
    invT = ops.inv_temperature(0.1)
    bufs = [make_batch(1000 * rank + 17 * i + 1, B, C, H, W, S, args.id_dtype, dev) for i in range(args.nbuf)]
    cls_w = torch.linspace(0.3, 1.0, C, device=dev)

    n_total = args.steps + args.warmup
    prob = torch.zeros((n_total, B, C), dtype=torch.int64, device=dev)
    csum = torch.zeros((n_total, B, S, C), dtype=torch.int64, device=dev)
    hist = torch.zeros((n_total, B, S, C), dtype=torch.int32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_total)]
    w31 = torch.from_numpy(ops.weights_to_fixed31(cls_w.cpu().numpy()).view(np.int32)).to(dev)

    def step(i):
        z, spx = bufs[i % args.nbuf]
        timed = i >= args.warmup and (i - args.warmup) % EVENT_EVERY == 0     # an event pair costs ~2 barrier packets: sample
        if timed:
            ev[i][0].record()
        ops.single_pass_accum(z, spx, S, invT, prob_sum=prob[i], class_sum=csum[i], hist=hist[i])
        if timed:
            ev[i][1].record()

    def finish(lo, hi):
        """Weighted means + ban for every region scored in steps [lo, hi), then ordering + budget walk."""
        n_img = (hi - lo) * B
        score = ops.region_finalize_weighted(csum[lo:hi].view(n_img, S, C), hist[lo:hi].view(n_img, S, C), w31, C - 1)[0]
        rank_t = torch.arange(n_img, dtype=torch.int32, device=dev)
        keys = ops.sort_keys_desc(ops.region_keys(score, None, rank_t))
        budget = max(1, int(100000 * n_img / 2975))
        nsel, simg, sid, ssc = ops.budget_walk(keys, None, rank_t, S, budget, budget + 1)
        return int(nsel.item())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # untimed clock ramp: the first ~50 launches after an idle period run ~20 % slower than the steady state (power
    # management), whatever --warmup the caller passes; the scan is re-run on the warm-up slots, results discarded
    for k in range(args.ramp):
        z, spx = bufs[k % args.nbuf]
        ops.single_pass_accum(z, spx, S, invT, prob_sum=prob[0], class_sum=csum[0], hist=hist[0])
    prob[0].zero_(); csum[0].zero_(); hist[0].zero_()
    for i in range(args.warmup):
        step(i)
    if args.warmup:
        finish(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        step(i)
    n_selected = finish(args.warmup, n_total)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    sampled = [ev[i] for i in range(args.warmup, n_total) if (i - args.warmup) % EVENT_EVERY == 0]
    sp_ms = float(np.mean([a.elapsed_time(b) for a, b in sampled]))
    id_bytes = {"int64": 8, "int32": 4, "int16": 2}[args.id_dtype]
    # algorithmic bytes of one k_single_pass launch: logits + ids read once, (prob + class sums + hist) written once
    sp_bytes = B * (C * H * W * 4 + H * W * id_bytes + S * C * (8 + 4) + C * 8)
    ach = sp_bytes / (sp_ms * 1e-3) / 1e9

    # reference-structured two-pass kernels on the same buffers (secondary; not part of `value`)
    def time_kernel(fn, n=100):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for k, (a, b) in enumerate(es):
            a.record(); fn(k); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in es]))
    p2 = torch.zeros((B, C), dtype=torch.int64, device=dev)
    s2 = torch.zeros((B, S), dtype=torch.int64, device=dev)
    h2 = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
    k2_ms = time_kernel(lambda k=0: ops.class_prob_sum(bufs[k % args.nbuf][0], invT, out=p2))
    k3_ms = time_kernel(lambda k=0: ops.bvsb_region_accum(bufs[k % args.nbuf][0], bufs[k % args.nbuf][1], cls_w, S, invT,
                                                          score_sum=s2, hist=h2))
    k2_bytes = B * (C * H * W * 4)
    k3_bytes = B * (C * H * W * 4 + H * W * id_bytes + S * (8 + 4 * C))
    default_shape = (B, C, H, W, S, args.id_dtype) == (4, 20, 1024, 2048, 2048, "int64")
    out = {
        "metric": "superpixels scored/sec", "value": args.steps * B * S * world / dt, "unit": "superpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "acquisition-scan: PixBal+ban-ignore scorer on resident logits, Cityscapes pool shape; per step one "
                               "single-pass scan (class prior + region sums + histograms); finalize + K4 selection once per run",
                   "images_per_step": B, "logits": [B, C, H, W], "nseg": S, "id_dtype": args.id_dtype,
                   "temperature": 0.1, "sharding": "pool images across ranks", "regions_selected": n_selected},
        "roofline": {"bound": "hbm", "kernel": "k_single_pass", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic("k_single_pass", default_shape),
                     "bytes_per_launch": sp_bytes, "avg_launch_ms": sp_ms},
        "two_pass": {"k_class_prob_sum": {"avg_launch_ms": k2_ms, "achieved_GBs": k2_bytes / (k2_ms * 1e-3) / 1e9,
                                          "bytes_per_launch": k2_bytes},
                     "k_bvsb_region_accum": {"avg_launch_ms": k3_ms, "achieved_GBs": k3_bytes / (k3_ms * 1e-3) / 1e9,
                                             "frac_of_peak": k3_bytes / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "bytes_per_launch": k3_bytes,
                                             "traffic": pmc_traffic("k_bvsb_region_accum", default_shape)},
                     "superpixels_per_s": B * S * world / ((k2_ms + k3_ms) * 1e-3)},
    }
    def secondary(fn, *a):
        """Secondary legs never take the primary line down with them."""
        try:
            return fn(*a)
        except Exception as e:          # noqa: BLE001
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    out["train_iter"] = None if args.no_train else secondary(train_iter_bench, args, dev, world)
    out["acquisition_with_model"] = None if args.no_train else secondary(acquisition_with_model_bench, args, dev, world)
    out["stage2"] = None if (args.no_train or rank != 0) else secondary(stage2_bench, args, dev)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
