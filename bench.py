#!/usr/bin/env python
"""bench.py -- MulActSeg hot path on MI355X: superpixels scored/sec (+ train-iter images/sec).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`, one rank per GPU;
  run bare (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script starts those N ranks itself as child
  processes and relays rank 0's line; a WORLD_SIZE that differs from --gpus is refused (exit 2).

Primary workload "acquisition-round" (BASELINE.json metric "superpixels scored/sec"; config 3's shapes).  One step = one
reference batch (val_batch_size = 4 pool pictures: logits [4,20,1024,2048] f32 + superpixel ids, resident in HBM) through
the single-pass scan of the PixBal + ban-ignore selector (class prior + per-region margin sums + arg-max histograms,
reference active_selection/my_bvsb_predclsbal_pwr_banignore.py:35-72).  Every rank scans K batches of ITS shard (weak
scaling: per-GPU work fixed); then, still inside the timed region, the round is finished exactly as the selector plugins
finish it (`engine.AcquisitionRound` -- the product object, not a private loop):
  exchange 1  all-gather of the per-picture class sums (RCCL)  ->  device f64 class weights (k_class_weight, no host round trip)
  weighted region means + ban (k_region_finalize_weighted) on the rank's rows
  exchange 2  all-gather of the region scores (RCCL)
  K4          64-bit keys + radix sort of the rank's own regions, all-gather of the per-rank heads (budget + 1 keys each), merged
              sort + fair-counting budget walk on every rank (engine.select_regions; one rank: keys, sort, walk over all regions).
value = N * K * 4 * 2048 / seconds (max over ranks, barrier + synchronize on both sides).

Secondary legs (same JSON line): the reference-structured two-pass kernels; "pool_round" -- the FIXED 2 975-picture x
2 048-superpixel pool with a 100 000-click budget through RegionSelector.select_next_batch, sharded over the ranks (strong
scaling), scan-only and with the model forward; "train_iter" / "train_iter_769" -- the stage-1 step at the reference's real
768x768 crop and at BASELINE.json's literal 769x769; "acquisition_with_model"; "stage2"; "cpu_baseline" (rank 0, N = 1).

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable
EVENT_EVERY = 4            # HIP-event pairs around every 4th scan launch of the timed region
POOL_IMAGES, POOL_CLICKS = 2975, 100000
POOL_WARM_IMAGES, POOL_WARM_CLICKS = 64, 2000


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
    ap.add_argument("--ramp", type=int, default=500, help="untimed launches before the warm-up steps (GPU clock ramp: 100 launches leave the scan ~6 % slower than 400+, tools/ramp_probe.sh)")
    ap.add_argument("--nbuf", type=int, default=3, help="distinct resident batches rotated through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the model legs (train-iter, acquisition with model, stage 2)")
    ap.add_argument("--no-trainleg", action="store_true", help="skip the train-iter and stage-2 legs but keep the acquisition leg with the model forward")
    ap.add_argument("--no-pool", action="store_true", help="skip the fixed-pool (strong scaling) acquisition round")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--acq-steps", type=int, default=16, help="steps of the secondary model-forward + scan measurement")
    ap.add_argument("--crop", type=int, default=768, help="training crop (reference: 768, transform.py:107)")
    ap.add_argument("--cpu-images", type=int, default=32, help="pictures of the CPU-baseline scorer sample (SURVEY 8(d): >= 32)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions at the reference's 20 threads (median reported; 8(d): 3)")
    ap.add_argument("--cpu-loss-steps", type=int, default=10, help="loss fwd+bwd steps on the host (8(d): 10)")
    ap.add_argument("--cpu-allcore-images", type=int, default=8,
                    help="pictures of the one extra scorer run at os.cpu_count() threads (torch-CPU is 4-6x slower at the box's 256 "
                         "hardware threads than at 20; reported, never the baseline unless it wins)")
    ap.add_argument("--cpu-full", action="store_true", help="kept for compatibility: the defaults are the 8(d) protocol now")
    return ap.parse_args()


def make_batch(seed, B, C, H, W, S, id_dtype, device):
    """Synthetic Cityscapes-shaped batch generated on the device (plumbing): cosine-like logits with a
    blocky class layout, jittered-grid superpixel map (mulactseg_amd.synth_pool)."""
    from mulactseg_amd.synth_pool import device_superpixel_maps
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    z = 0.35 * torch.randn((B, C, H, W), generator=g, device=device, dtype=torch.float32)
    cm = torch.randint(0, C, (B, 1, H // 32 + 1, W // 32 + 1), generator=g, device=device)
    cm = cm.repeat_interleave(32, 2).repeat_interleave(32, 3)[:, :, :H, :W]
    z.scatter_add_(1, cm, torch.full_like(cm, 0.6, dtype=torch.float32))
    spx = device_superpixel_maps([seed * 131 + i for i in range(B)], H, W, S, device, torch.int16)
    return z.contiguous(), spx.to(getattr(torch, id_dtype)).contiguous()


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def device_sync():
    """Drain the GPU; nothing to do when the host logic of a leg is rehearsed on the CPU (tests/test_bench_world2_cpu.py: gloo,
    the oracle-backed stand-in for the HIP backend)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class _NoEvent:
    """torch.cuda.Event for a CPU rehearsal: records nothing."""

    def record(self, *a):
        pass

    def elapsed_time(self, other):
        return 0.0


def timing_event(dev):
    return torch.cuda.Event(enable_timing=True) if torch.device(dev).type == 'cuda' else _NoEvent()


def fence():
    device_sync()
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.barrier()
        device_sync()


def max_over_ranks(seconds, dev):
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------------------------
# primary leg
# ------------------------------------------------------------------------------------------------------------------
class ScanRound:
    """K batches of this rank's shard through engine.AcquisitionRound, then the round's tail (two exchanges, device class
    weights (k_class_weight), weighted finalize + ban, sharded K4 with a fair-counting budget)."""

    def __init__(self, args, dev, rank, world, backend, bufs, n_steps, cost_seed=5):
        from mulactseg_amd.active_selection.engine import AcquisitionRound
        B, C, S = args.batch, args.classes, args.nseg
        self.args, self.bufs, self.n_steps, self.backend = args, bufs, n_steps, backend
        self.n_img = world * n_steps * B
        self.rnd = AcquisitionRound(self.n_img, C, S, B, 0.1, backend, rank=rank, world=world, single_pass=True)
        assert self.rnd.plan.n_local == n_steps * B
        g = torch.Generator(device=dev)
        g.manual_seed(cost_seed)
        # click cost per region under fair counting + or-labeling = number of classes under it (1..4), replicated
        self.cost = (torch.rand((self.n_img, S), generator=g, device=dev) ** 3 * 4).to(torch.uint8).clamp_(0, 3) + 1
        self.rank_t = torch.arange(self.n_img, dtype=torch.int32, device=dev)
        self.budget = max(1, int(POOL_CLICKS * self.n_img / POOL_IMAGES))
        self.events = []
        self.dev = dev
        self.tail_ev = (timing_event(dev), timing_event(dev))

    def scan(self, i, timed):
        z, spx = self.bufs[i % len(self.bufs)]
        if timed:
            a, b = timing_event(self.dev), timing_event(self.dev)
            a.record()
        self.rnd.add_single_pass(i * self.args.batch, z, spx)
        if timed:
            b.record()
            self.events.append((a, b))

    def tail(self):
        C = self.args.classes
        self.tail_ev[0].record()
        cls_w = self.rnd.class_weights(6.0)                                    # exchange 1 + k_class_weight (device f64)
        scores = self.rnd.scores_single_pass(cls_w, ban_class=C - 1)           # finalize + ban, exchange 2
        from mulactseg_amd.active_selection.engine import select_regions
        n, simg, sid, ssc = select_regions(self.backend, self.rnd.plan, scores, None, self.rank_t, self.rank_t, self.cost, self.budget,
                                           self.budget + 1)
        self.tail_ev[1].record()
        self.scores = scores
        self.selected = (simg, sid, ssc)
        return n

    def run(self, timed):
        for i in range(self.n_steps):
            self.scan(i, timed and i % EVENT_EVERY == 0)
        return self.tail()


def timed_scan_round(args, dev, rank, world, backend, bufs, ramp=None):
    """The primary leg: W warm-up batches, then EXACTLY K timed batches of this rank's shard + the round's tail, bracketed by a
    barrier + synchronize on both sides; the time is the MAX over ranks.  -> (the timed ScanRound, regions selected, seconds).
    (tests/test_bench_world2_cpu.py runs this function under gloo with world 2 against world 1.)"""
    if args.warmup:
        ScanRound(args, dev, rank, world, backend, bufs, args.warmup).run(False)
    timed = ScanRound(args, dev, rank, world, backend, bufs, args.steps)
    timed.rnd.hw = args.height * args.width
    timed.tail()            # untimed, on the still-empty accumulators: the timed round's buffers come out of torch's caching
    #                         allocator instead of hipMalloc (a long-lived trainer process is in that state from round 2 on)
    # untimed clock ramp, LAST thing before the timed region: the first ~50 launches after an idle period run ~20 % slower than the
    # steady state (power management), and the set-up above (allocations, the warm-up round's tail with its host read) leaves the
    # GPU idle for milliseconds.  Until round 4 the ramp ran BEFORE that set-up: the driver's 20 timed steps (3.2 ms) then started
    # on a GPU that had just idled and read 166 us per scan where 200 steps read 153 (same code, same box).
    if ramp is not None:
        ramp()
    fence()
    t0 = time.perf_counter()
    n_selected = timed.run(True)
    fence()
    return timed, n_selected, max_over_ranks(time.perf_counter() - t0, dev)


# ------------------------------------------------------------------------------------------------------------------
# fixed-pool round (strong scaling) through the selector plugin
# ------------------------------------------------------------------------------------------------------------------
class ModelOnRotatingPictures(torch.nn.Module):
    """trainer.net for the pool round WITH the model forward: picture indices in, logits of the real network out.  The
    2 975 x 25 MB of pool pictures are rotated out of `nbuf` resident synthetic batches (the forward's cost does not
    depend on the pixel values)."""

    def __init__(self, net, B, H, W, dev, nbuf=2):
        super().__init__()
        self.net = net
        g = torch.Generator(device=dev)
        g.manual_seed(31)
        self.pictures = torch.randn((nbuf, B, 3, H, W), generator=g, device=dev)
        self.k = 0

    def forward(self, indices):
        self.k += 1
        return self.net(self.pictures[self.k % self.pictures.shape[0]][:indices.shape[0]])


def pool_round_bench(args, dev, rank, world, with_model, n_images=POOL_IMAGES, clicks=POOL_CLICKS, backend=None, save_dir=None):
    """BASELINE.json config 3 at its real size: 2 975 pictures x 2 048 superpixels, 100 000 clicks (fair counting), through
    RegionSelector.select_next_batch -- scan (+ model forward), two exchanges, device class weights, finalize + ban, K4 on
    6.09 M keys, RegionActiveDataset bookkeeping.  The pool is sharded over the ranks: strong scaling.
    (n_images / clicks: the warm-up round on a small pool that precedes the timed one.  backend / save_dir: the CPU rehearsal of this
    function's host logic under gloo, tests/test_bench_world2_cpu.py -- the selector's HIP backend replaced by the oracle-backed
    stand-in, the run directory named by the caller so that it can read the selection pickle.)"""
    from mulactseg_amd.active_selection import my_bvsb_predclsbal_pwr_banignore as banignore
    from mulactseg_amd.active_selection.engine import ShardPlan
    from mulactseg_amd.dataloader import RegionActiveDataset
    from mulactseg_amd.synth_pool import LogitSource, SyntheticLabels, SyntheticPool
    B, C, H, W, S = args.batch, args.classes, args.height, args.width, args.nseg
    plan = ShardPlan(n_images, B, rank, world)
    pool = SyntheticPool(n_images, H, W, S, dev, shard=(plan.img_lo, plan.img_hi))
    labels = SyntheticLabels(pool, C)
    if with_model:
        from mulactseg_amd.models import get_model
        net = ModelOnRotatingPictures(get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).eval(),
                                      B, H, W, dev)
    else:
        net = LogitSource(C, H, W, dev, nbuf=B + 2, window=(B, plan.batch_lo))        # three distinct resident batches, zero-copy
    tmp = save_dir if save_dir is not None else tempfile.mkdtemp(prefix="mas_pool_r%d_" % rank)
    a = types.SimpleNamespace(val_batch_size=B, val_num_workers=0, nseg=S, active_method='pixbal', num_classes=C - 1, ce_temp=0.1,
                              cls_weight_coeff=6.0, method='active_joint_multi_predignore_lossdecomp', save_scores=False,
                              fair_counting=True, or_labeling=True, model_save_dir=tmp, finetune_itrs=1,
                              wandb=types.SimpleNamespace(log=lambda *x, **k: None))
    marks = {}

    class Timed(banignore.RegionSelector):
        def calculate_scores_tensor(self, trainer, pool_set, want_hist=False):
            out = super().calculate_scores_tensor(trainer, pool_set, want_hist)
            device_sync()
            marks['scored'] = time.perf_counter()
            return out

    sel = Timed(a)
    if backend is not None:
        sel.backend = backend
    trainer = types.SimpleNamespace(net=net, device=dev, model_save_dir=tmp, selection_iter=1)
    active = RegionActiveDataset(a, pool, labels)
    active.selection_iter = 1
    expand = active.expand_training_set

    def timed_expand(*x, **k):
        device_sync()
        marks['selected'] = time.perf_counter()
        return expand(*x, **k)
    active.expand_training_set = timed_expand
    if with_model:                              # MIOpen's find runs on the first calls of a new shape
        with torch.no_grad():
            for _ in range(3):
                net(torch.zeros(B))
    fence()
    t0 = time.perf_counter()
    sel.select_next_batch(trainer, active, clicks)
    fence()
    t1 = time.perf_counter()
    active.wait_for_writes()        # (the selection pickle is written by a background thread, off the round's critical path)
    t_write = time.perf_counter() - t1
    dt = max_over_ranks(t1 - t0, dev)
    # what follows the round in the reference loop (train_AL.py:69): the datalist pickle of all four lists.  The pool / label id
    # lists are held as arrays until somebody reads them (dataloader/region_active_dataset.py:LazySuppix); this call reads all of
    # them, so the Python lists the round did not build are built HERE -- reported beside the round, not hidden in it.
    pending = getattr(pool.suppix, 'pending', lambda: 0)()
    t2 = time.perf_counter()
    active.dump_datalist()
    t_dump = time.perf_counter() - t2
    n_sel = sum(len(v) for v in labels.suppix.values())
    return {"seconds": dt, "superpixels_per_s": n_images * S / dt, "regions_selected": n_sel,
            "rank0_breakdown_s": {"scores (scan%s + exchanges + class weights + finalize)" % (" + model forward" if with_model else ""):
                                  marks['scored'] - t0,
                                  "valid mask + cost table + K4 (keys, radix sort, walk)": marks['selected'] - marks['scored'],
                                  "RegionActiveDataset.expand_training_set (host; the selection pickle is written by a background thread)": t1 - marks['selected'],
                                  "selection pickle still being written after the round returned": t_write},
            "after_the_round_s": {"RegionActiveDataset.dump_datalist (train_AL.py:69; rank 0; builds the %d id lists the round kept as arrays, "
                                  "pickles 6.09 M ids)" % pending: t_dump},
            "images_per_rank": plan.n_local}


# ------------------------------------------------------------------------------------------------------------------
# model legs
# ------------------------------------------------------------------------------------------------------------------
LOSS_PMC_BYTES = (61.1 + 280.6) * 1e6      # k_partial_loss_fwd + k_partial_loss_bwd at [4,20,768,768], 8 % selected: profiles/r06/n_loss_hbm_pmc.md
F32_MFMA_PEAK_TF = 157.3        # dense f32 MFMA peak of MI355X (MI355X_MICROARCH.md), TFLOP/s
# The convolutions that run on csrc/conv_bx.hip / conv_wgrad_bx.hip compute the same f32 products from exact three-term bf16 splits
# of both operands: six bf16 MFMAs (16x the f32 rate) per sixteen f32 ones -- the matrix-core bound of an f32 convolution done that
# way is 16 / 6 x the f32 peak.  A leg whose layers run (almost) all in that form is priced against THAT bound (`peak_TFLOPs`,
# `mfma_frac` <= 1); `f32_pipe_peak_TFLOPs` is printed beside it for reference only -- the f32 pipe alone could not reach the
# achieved rate, so no fraction is formed against it.
SPLIT_BF16_BOUND_TF = F32_MFMA_PEAK_TF * 16.0 / 6.0
# (rounds 1-4 printed `mfma_frac` against the f32 peak; since round 5 it is the fraction of the split-bf16 bound.  Both ratios are printed
#  under names that say which: compare `frac_of_f32_mfma_peak` with the rounds-1-4 `mfma_frac`.  The f32-pipe ratio may exceed 1.)
MFMA_FRAC_IS = "frac_of_split_bf16_bound (since r05; r01-r04 lines quoted frac_of_f32_mfma_peak under this key)"
MFMA_ARITH = ("f32 operands and f32 accumulation; layers marked hip_bx / '/bx' in layer_paths_per_step and the 1x1 / 3x3 stride-1 weight gradients run on "
              "v_mfma_f32_32x32x16_bf16 from exact three-term bf16 splits of both operands (six partial products, dropped terms <= 2^-23 "
              "of a product: csrc/bx_split.h), the others on v_mfma_f32_32x32x2_f32")


def conv_flop(net, N, H, W, products):
    """FLOPs of the dense (groups == 1) convolutions of `net` on a [N,3,H,W] batch, counted with forward hooks on one CPU-free
    shape walk: 2 * taps * Cin * Cout * output pixels per product; `products` = 1 (inference) or 3 (training: the first layer has
    no input gradient)."""
    total = [0.0]
    first = [True]
    hooks = []

    def hook(m, inp, out):
        if m.groups != 1:
            return
        f = 2.0 * m.kernel_size[0] * m.kernel_size[1] * m.in_channels * m.out_channels * out.shape[0] * out.shape[2] * out.shape[3]
        total[0] += f * (products - 1 if (first[0] and products == 3) else products)
        first[0] = False
    # a meta-device walk costs nothing and takes no kernel path of the package (plain torch ops on shapes only)
    import copy
    meta = copy.deepcopy(net).to('meta').eval()
    hooks = [m.register_forward_hook(hook) for m in meta.modules() if isinstance(m, torch.nn.Conv2d)]
    with torch.no_grad():
        meta(torch.empty((N, 3, H, W), device='meta'))
    return total[0]


def train_step_flop(net, N, crop):
    return conv_flop(net, N, crop, crop, 3)


def train_iter_bench(args, dev, world, crop):
    """Secondary metric "train-iter images/sec" (BASELINE.json configs[1]): stage-1 step on a
    [4,20,crop,crop] batch.  (a) loss-only: fused partial-label losses fwd+bwd on resident logits;
    (b) full iteration: DeepLabv3+WN/ResNet50-deepstem fwd + losses + bwd + AdamW (fp32, random init)."""
    from mulactseg_amd import synth
    from mulactseg_amd.models import deeplab, get_model
    from mulactseg_amd.utils.loss import FusedPartialLabelLoss
    N, C, S = 4, args.classes, args.nseg
    spx, msk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    spx = torch.from_numpy(np.stack(spx)).to(dev)
    msk = torch.from_numpy(np.stack(msk)).to(dev)
    tgt = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)])).to(dev)
    crit = FusedPartialLabelLoss(S, 0.1, 0.1, sync_normalisers=True)     # global 1 + n over the data-parallel batch
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

    # the production form: quarter-resolution logits in, the x4 bilinear upsampling evaluated inside the scans (no [N,C,H,W]
    # logit / gradient tensors); compared with upsample + loss + both backwards of the materialised path
    from mulactseg_amd import ops
    q = ((crop - 1) // 2) // 2 + 1
    zq = (0.35 * torch.randn((N, C, q, q), generator=g, device=dev)).requires_grad_(True)

    def lowres_step(logits_q):          # the production trainer's call: objective and chain rule inside the loss kernels
        total, _, _, _ = crit.weighted_lowres(logits_q, (crop, crop), tgt, spx, msk, 16.0, 8.0, 1.0)
        return total

    def timed_loss(fn, leaf, n=20):
        for _ in range(3):
            leaf.grad = None
            fn().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            leaf.grad = None
            fn().backward()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    low_ms = timed_loss(lambda: lowres_step(zq), zq)
    mat_ms = timed_loss(lambda: loss_step(ops.upsample_bilinear(zq, (crop, crop))), zq)

    # The same two directions as the GPU runs them, without the host in the measurement: the fused entry points (one library call
    # per direction: prep + scan + finalize-with-values / memset + scan + conversion) issued back to back, timed with HIP events on
    # their stream.  The wall-clock figures above are bound by the host (autograd bookkeeping, the torch arithmetic that composes
    # the objective) whenever nothing else keeps the GPU busy; inside a training step these launches queue behind the model's.
    invT, wts = ops.inv_temperature(0.1), torch.tensor([16.0, 8.0, 1.0], device=dev)
    go3, go1 = torch.tensor([16.0, 8.0, 1.0], device=dev), torch.ones(1, device=dev)

    def gpu_time(fn, n=30):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def full_raw():
        _, st8 = ops.partial_loss_fwd_fused(z.detach(), None, spx, msk, invT, crit.flags, targets=tgt)
        ops.partial_loss_bwd_fused(z.detach(), None, spx, msk, st8, go3, invT)

    def low_raw():
        _, st8 = ops.partial_loss_fwd_fused(zq.detach(), (crop, crop), spx, msk, invT, crit.flags, targets=tgt, weights=wts)
        ops.partial_loss_bwd_fused(zq.detach(), (crop, crop), spx, msk, st8, go1, invT, weights=wts)
    loss_gpu_ms, low_gpu_ms = gpu_time(full_raw), gpu_time(low_raw)

    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).to(dev).train()
    groups = [{'params': list(net.backbone.parameters()), 'lr': 2e-5}, {'params': list(net.classifier.parameters()), 'lr': 2e-4}]
    if os.environ.get("MAS_ADAMW", "own") == "own":       # what trainer/base.py:get_optim builds on the GPU (csrc/optim.hip: one launch)
        from mulactseg_amd.utils.optim import FusedAdamW
        opt = FusedAdamW(groups, lr=2e-5, weight_decay=1e-5)
    else:
        opt = torch.optim.AdamW(groups, lr=2e-5, weight_decay=1e-5, fused=True)
    if _dist() is not None:       # data parallel as the trainers run it: gradients all-reduced over RCCL
        net = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index], output_device=dev.index)
    images = torch.randn((N, 3, crop, crop), generator=g, device=dev)

    def full_step():
        opt.zero_grad(set_to_none=True)
        (lowres_step(net(images, lowres=True)) * world).backward()          # as the production trainer's train_impl does
        opt.step()

    for _ in range(5):                  # (the leg starts from an empty caching allocator: it settles within these steps)
        full_step()
    deeplab.path_report(reset=True)
    full_step()
    paths = deeplab.path_report(reset=True)
    fence()
    mallocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        full_step()
    fence()
    it_ms = max_over_ranks(time.perf_counter() - t0, dev) / args.train_steps * 1e3
    mallocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mallocs0
    sk_err = ops.conv_sk_error(dev)         # (synchronises; the trainers read the same word with the loss)
    if sk_err != 0:
        raise ops.StreamKGaveUp("bench train leg: a stream-K convolution gave up (error word %d): the timed steps are invalid" % sk_err)
    flop = train_step_flop(net.module if hasattr(net, "module") else net, N, crop)
    return {"metric": "train-iter images/sec", "value": N * world / (it_ms * 1e-3), "unit": "images/s", "ms_per_iter": it_ms,
            "stream_k_error_word": sk_err, "timed_steps": args.train_steps,
            "hipMalloc_calls_in_the_timed_steps": mallocs, "reserved_GB": torch.cuda.memory_reserved(dev) / 2 ** 30,
            "weight_gradient_stream": os.environ.get("MAS_WGRAD_STREAM", "async") if _dist() is None else "main (torch.distributed)",
            "mfma": {"flop_per_step": flop, "achieved_TFLOPs": flop / (it_ms * 1e-3) / 1e12, "peak_TFLOPs": SPLIT_BF16_BOUND_TF,
                     "mfma_frac": flop / (it_ms * 1e-3) / 1e12 / SPLIT_BF16_BOUND_TF,
                     "frac_of_split_bf16_bound": flop / (it_ms * 1e-3) / 1e12 / SPLIT_BF16_BOUND_TF,
                     "frac_of_f32_mfma_peak": flop / (it_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF,
                     "mfma_frac_is": MFMA_FRAC_IS,
                     "peak_is": "the matrix-core bound of f32 convolutions computed from three-term bf16 splits (16/6 x the f32 MFMA peak)",
                     "f32_pipe_peak_TFLOPs": F32_MFMA_PEAK_TF,
                     "arithmetic": MFMA_ARITH,
                     "note": "2 * taps * Cin * Cout * output pixels of every dense convolution x 3 products (forward, input gradient, "
                             "weight gradient; no input gradient for the first layer), over the WHOLE step's wall time"},
            "config": {"workload": "stage-1 step: DeepLabv3+WN/ResNet50-deepstem fwd+bwd with the dense convolutions on this package's matrix-core "
                                   "kernels (k_conv_bx: forward / input gradient of every stride-1 layer from three-term bf16 splits of the f32 operands, the K chunks of "
                                   "the small planes' layers dealt to several workgroups; k_wgrad_bx / k_wgrad_bx3: 1x1 / 3x3 weight gradient on the same arithmetic "
                                   "(the 3x3 X patch through ds_read_b64_tr_b16); k_conv_sk / k_wgrad: persistent stream-K f32 forward / input gradient and "
                                   "split-K weight gradient of the stride-2 layers and the narrow 1x1 layers; layer_paths_per_step says which product of which "
                                   "layer took which kernel) + HIP memory-bound layers + "
                                   "fused partial-label losses (HIP) + AdamW" + ("; DistributedDataParallel over RCCL, global loss normalisers" if world > 1 else ""),
                       "train_conv_mode": os.environ.get("MAS_TRAIN_CONV", "own"),
                       "batch": [N, 3, crop, crop], "logits": [N, C, crop, crop], "nseg": S,
                       "selected_fraction": float(msk.float().mean())},
            "layer_paths_per_step": paths,
            "loss_only": {"ms_fwd_bwd": loss_ms, "algorithmic_GBs": loss_bytes / (loss_ms * 1e-3) / 1e9,
                          "bytes": loss_bytes,
                          "gpu_ms_fwd_bwd": loss_gpu_ms, "gpu_algorithmic_GBs": loss_bytes / (loss_gpu_ms * 1e-3) / 1e9,
                          "gpu_frac_of_hbm_peak": loss_bytes / (loss_gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          # the bytes the two scans really move (PMC, profiles/r06/n_loss_hbm_pmc.md at this shape and selected fraction: forward
                          # 61.1 MB -- it reads the mask and the logits of the SELECTED pixels only --, backward 280.6 MB -- it writes dz
                          # completely): the figure above prices the kernels on SURVEY section 8(d)'s full-tensor bytes, this one on the traffic
                          "pmc_bytes_fwd_bwd": LOSS_PMC_BYTES if crop == 768 else None,
                          "gpu_frac_of_hbm_peak_on_pmc_bytes": (LOSS_PMC_BYTES / (loss_gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if crop == 768 else None,
                          "note": "full-resolution logits as the leaf.  ms_fwd_bwd: wall clock of the loss modules through autograd, the objective "
                                  "composed with torch arithmetic (host-bound when the GPU has nothing else to do); gpu_ms_fwd_bwd: the same two "
                                  "directions through the fused entry points, HIP events (prep + forward scan + finalize, backward scan); "
                                  "per-kernel times and PMC bytes: profiles/r06/n_loss_*.md"},
            "loss_from_quarter_logits": {"ms_fwd_bwd_fused": low_ms, "gpu_ms_fwd_bwd_fused": low_gpu_ms,
                                         "ms_fwd_bwd_upsample_then_loss": mat_ms, "quarter_logits": [N, C, q, q],
                                         "note": "leaf = the model's quarter-resolution logits: fused = bilinear x4 evaluated inside both scans "
                                                 "(train_iter uses this); the other = upsample kernel + loss scans + dz + upsample backward"}}


def acquisition_with_model_bench(args, dev, world):
    """the acquisition metric "also with the model forwards".  One step = eval-mode forward of
    DeepLabv3+WN/ResNet50-deepstem (fp32, random init) on a [B,3,H,W] pool batch + the single-pass scan of its logits.
    The reference structure needs two forwards per image (class prior, then scores); the single-pass scan needs one."""
    from mulactseg_amd import ops
    from mulactseg_amd.models import deeplab, get_model
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

    from mulactseg_amd.active_selection.my_bvsb import pool_streams
    streams = pool_streams(dev)         # consecutive pool batches alternate between two streams, as RegionSelector._iterate runs them
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    count = [0]

    def step():
        # as the PixBal selectors run it: quarter-resolution logits out of the model, the final x4 bilinear upsampling
        # (models/segmentation/utils.py:25) evaluated inside the scan -- bit-identical to scanning the upsampled tensor
        ctx = torch.cuda.stream(streams[count[0] % len(streams)]) if streams else contextlib.nullcontext()
        count[0] += 1
        with ctx, torch.no_grad():
            zq = net(images, lowres=True)
            ops.single_pass_accum_lowres(zq.contiguous(), (H, W), spx, S, invT, prob_sum=prob, class_sum=csum, hist=hist)

    for _ in range(4):              # MIOpen's find runs on the first calls of every new shape
        step()
    torch.cuda.synchronize()
    deeplab.path_report(reset=True)
    step()
    paths = deeplab.path_report(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.acq_steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.acq_steps * 1e3
    # the two forms of the scan on the same quarter-resolution logits, kernel time by HIP events
    with torch.no_grad():
        zq = net(images, lowres=True).contiguous()

        def ev(fn, n=20):
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        low_ms = ev(lambda: ops.single_pass_accum_lowres(zq, (H, W), spx, S, invT, prob_sum=prob, class_sum=csum, hist=hist))
        mat_ms = ev(lambda: ops.single_pass_accum(ops.upsample_bilinear(zq, (H, W)), spx, S, invT, prob_sum=prob, class_sum=csum, hist=hist))
    scan_forms = {"scan_from_quarter_logits_ms": low_ms, "upsample_then_scan_ms": mat_ms,
                  "note": "k_single_pass<LOWRES> (interpolation in registers; bound by VALU work and LDS-read latency, "
                          "profiles/r02/k_scan_forms_pmc.md) vs k_upsample_fwd + k_single_pass_ring (671 MB written and re-read)"}
    flop = conv_flop(net, B, H, W, 1)
    return {"metric": "superpixels scored/sec incl. model forward", "scan_forms": scan_forms, "value": B * S * world / (ms * 1e-3), "unit": "superpixels/s",
            "ms_per_batch": ms, "forwards_per_image": 1, "layer_paths_per_step": paths,
            "mfma": {"flop_per_batch": flop, "achieved_TFLOPs": flop / (ms * 1e-3) / 1e12, "peak_TFLOPs": SPLIT_BF16_BOUND_TF,
                     "mfma_frac": flop / (ms * 1e-3) / 1e12 / SPLIT_BF16_BOUND_TF,
                     "frac_of_split_bf16_bound": flop / (ms * 1e-3) / 1e12 / SPLIT_BF16_BOUND_TF,
                     "frac_of_f32_mfma_peak": flop / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF,
                     "mfma_frac_is": MFMA_FRAC_IS,
                     "peak_is": "the matrix-core bound of f32 convolutions computed from three-term bf16 splits (16/6 x the f32 MFMA peak)",
                     "f32_pipe_peak_TFLOPs": F32_MFMA_PEAK_TF,
                     "arithmetic": MFMA_ARITH,
                     "note": "2 * taps * Cin * Cout * output pixels of every dense convolution of one forward, over the whole batch's wall time"},
            "config": {"workload": "eval forward (matrix-core convolutions with BatchNorm / residual / ReLU epilogues -- hip_bx: f32 products from three-term "
                                   "bf16 splits, hip_mfma: f32 MFMA -- + HIP memory-bound layers, no MIOpen kernel: see layer_paths_per_step) of [%d,3,%d,%d] + single-pass scan of the quarter-resolution "
                                   "logits; the reference structure runs the forward twice per pool image; consecutive batches alternate between %d HIP streams, as RegionSelector._iterate runs the pool (MAS_POOL_STREAMS)" % (B, H, W, max(1, len(streams)))}}


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
    res["generation_loop"] = stage2_loop(net, image, spx, tgt, msk, C, dev)
    return {"metric": "stage-2 pseudo-label generation, ms per 1024x2048 image (model forward + K9 kernels)", **res}


def stage2_loop(net, image, spx, tgt, msk, C, dev, pictures=32):
    """The LOOP the product runs over the labelled pictures (trainer/eval_save_cosplbl_prop_includeonehot.inference: forward at batch 1,
    K9 kernels, IoU counters, one PNG per picture) on `pictures` copies of the leg's synthetic picture: wall time per picture with the
    pictures dealt to MAS_STAGE2_WORKERS threads on their own streams (the default) and one picture at a time (rounds 1-5, the reference)."""
    import shutil
    from mulactseg_amd.trainer import eval_save_cosplbl_prop_includeonehot as G
    labels = torch.zeros((1,) + tuple(spx.shape[-2:]), dtype=torch.long, device=dev)

    class Loader:
        def __init__(self, n):
            self.n, self.k = n, 0

        def __len__(self):
            return self.n

        def __next__(self):
            self.k += 1
            return {'images': image, 'labels': labels, 'spx': spx, 'spmask': msk, 'target': tgt,
                    'fnames': [["i/p%05d.png" % self.k, "l/p%05d.png" % self.k, "s/p%05d.pkl" % self.k]]}
    out = {}
    keep = os.environ.get("MAS_STAGE2_WORKERS")
    try:
        for name, workers, n in (("ms_per_picture", keep or "4", pictures), ("ms_per_picture_one_at_a_time", "1", pictures // 2)):
            os.environ["MAS_STAGE2_WORKERS"] = workers
            tmp = tempfile.mkdtemp(prefix="mas_s2loop_")
            tr = object.__new__(G.ActiveTrainer)
            tr.args = types.SimpleNamespace(ignore_idx=255, init_checkpoint=os.path.join(tmp, "checkpoint01.tar"), plbl_type=None, val_batch_size=1)
            tr.net, tr.device, tr.num_classes, tr.selection_iter, tr.save_dir = net, dev, C - 1, 1, None
            with contextlib.redirect_stdout(sys.stderr):            # (the trainer prints its IoU table)
                tr.inference(Loader(6))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tr.inference(Loader(n))
                torch.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) / n * 1e3
            shutil.rmtree(tmp, ignore_errors=True)
        out["worker_threads"] = int(keep or "4")
    finally:
        if keep is None:
            os.environ.pop("MAS_STAGE2_WORKERS", None)
        else:
            os.environ["MAS_STAGE2_WORKERS"] = keep
    return out


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): oracle/port.py = the reference's op sequence in torch-CPU, pinned bit-exact to the
# executed reference by tests/test_oracle_golden.py.  This is the ONLY place bench.py touches oracle/.
# ------------------------------------------------------------------------------------------------------------------
def numpy_select(scores, cost, budget):
    """sorted(tuples, reverse=True) + budget walk on arrays (path rank = picture index): consumed (picture, id) pairs."""
    n, s = scores.shape
    img = np.repeat(np.arange(n), s)
    rid = np.tile(np.arange(s), n)
    order = np.lexsort((-rid, -img, -scores.reshape(-1).astype(np.float64)))
    cum = np.cumsum(cost.reshape(-1)[order].astype(np.int64))
    over = np.nonzero(cum > budget)[0]
    m = len(order) if len(over) == 0 else int(over[0]) + 1
    return img[order[:m]], rid[order[:m]]


def cpu_baseline(args, dev, bufs, backend):
    """the reference CPU path on the GPU box's host cores, on the SAME tensors the GPU leg scans (the
    resident batches, copied to the host).  Scorer round (both passes + ban + tuple list + Python sort + budget walk) at
    os.cpu_count() threads and at the reference's own default of 20 (utils/common.py:343), the partial-label losses
    fwd+bwd on [4,20,768,768], and the model forward on one pool batch.  Defaults = the protocol's sizes (32 pictures, median
    of 3 at 20 threads, 10 loss steps: ~40 s); the run at every hardware thread is one repetition on --cpu-allcore-images."""
    from mulactseg_amd import synth
    from oracle import port
    cores = os.cpu_count() or 1
    B, C, H, W, S = args.batch, args.classes, args.height, args.width, args.nseg
    n_img = max(B, args.cpu_images // B * B)
    reps = max(1, args.cpu_reps)
    loss_steps = max(1, args.cpu_loss_steps)
    n_b = n_img // B
    host = [(z.cpu(), spx.cpu().to(torch.int64)) for z, spx in bufs]                  # generated once on the device, copied
    im_idx = [["img_%05d.png" % i, "lbl_%05d.png" % i, "spx_%05d.pkl" % i] for i in range(n_img)]
    suppix = {k[2]: list(range(S)) for k in im_idx}
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    cost = ((torch.rand((n_img, S), generator=g, device=dev) ** 3 * 4).to(torch.uint8).clamp_(0, 3) + 1).cpu().numpy()
    budget = max(1, int(POOL_CLICKS * n_img / POOL_IMAGES))
    row = {','.join(k): i for i, k in enumerate(im_idx)}

    def scorer_round(n_pictures=None):
        """calculate_scores (pass 1, class weights, pass 2, ban) -> tuple list -> sorted(reverse=True) -> budget walk"""
        n_p = n_img if n_pictures is None else n_pictures
        nb = n_p // B
        means = [port.class_prior_batch(host[b % len(host)][0], 0.1) for b in range(nb)]
        _, w = port.class_weight(means, 6.0)
        rb, rh = [], []
        for b in range(nb):
            z, spx = host[b % len(host)]
            r, h = port.region_scores_batch(z, spx, 0.1, w, S, C)
            rb.append(r)
            rh.append(h)
        sc, _ = port.ban_ignore_dominant(torch.cat(rb).view(-1), torch.cat(rh).view(-1, C))
        sc = sc.view(n_p, S)
        tuples = port.score_list(im_idx[:n_p], suppix, sc)
        taken = port.select_regions(tuples, max(1, int(POOL_CLICKS * n_p / POOL_IMAGES)), lambda path, rid: int(cost[row[path], rid]))
        return sc.numpy(), taken

    def timed(threads, n_rep, n_pictures=None):
        torch.set_num_threads(threads)
        ts = []
        for _ in range(n_rep):
            t0 = time.perf_counter()
            out = scorer_round(n_pictures)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts, out

    # the protocol's run: the reference's own default thread count (utils/common.py:343), n_img pictures, median of `reps`
    t_20, ts_20, (sc_cpu, taken) = timed(min(20, cores), reps)
    # one extra run at every hardware thread of the box on a smaller sample, scaled to n_img pictures for the comparison
    n_all = max(B, min(n_img, args.cpu_allcore_images // B * B))
    if cores > 20:
        t_all_raw, ts_all, _ = timed(cores, 1, n_all)
        t_all = t_all_raw * n_img / n_all
    else:
        t_all, ts_all = t_20, ts_20
    best = min(t_all, t_20)

    # the same pictures through the GPU round: selected sets compared (the reference's f32 order vs the fixed-point scan)
    from mulactseg_amd.active_selection.engine import AcquisitionRound
    rnd = AcquisitionRound(n_img, C, S, B, 0.1, backend, rank=0, world=1, single_pass=True)
    for b in range(n_b):
        z, spx = bufs[b % len(bufs)]
        rnd.add_single_pass(b * B, z, spx)
    sc_gpu = rnd.scores_single_pass(rnd.class_weights(6.0), ban_class=C - 1).cpu().numpy()
    gi, gid = numpy_select(sc_gpu, cost, budget)
    gpu_set = set(zip(gi.tolist(), gid.tolist()))
    cpu_set = {(row[p], r) for _, p, r in taken}
    nz = sc_cpu != 0
    sel_cmp = {"selected_gpu": len(gpu_set), "selected_cpu": len(cpu_set), "symmetric_difference": len(gpu_set ^ cpu_set),
               "max_rel_delta_score": float((np.abs(sc_gpu[nz] - sc_cpu[nz]) / sc_cpu[nz]).max()),
               "same_zero_regions": bool(np.array_equal(nz, sc_gpu != 0))}

    # losses: OnehotCEMultihotChoice + GroupMultiLabelCE_onlymulti fwd + bwd (utils/loss.py:81-141,535-588 subclasses).
    # Losses and model forward run at the reference's own default thread count (utils/common.py:343: 20): on the GPU box's
    # 256 hardware threads torch-CPU is 4-6x SLOWER than at 20 (measured r02: 51.8 s vs ~8 s per loss step), and the
    # baseline must not be flattered down.
    lm_threads = min(20, cores)
    torch.set_num_threads(lm_threads)
    N, crop = 4, 768
    sp, mk = zip(*[synth.train_crop(50 + i, crop, crop, S, frac_selected=0.09) for i in range(N)])
    sp, mk = torch.from_numpy(np.stack(sp)), torch.from_numpy(np.stack(mk))
    tg = torch.from_numpy(np.stack([synth.multi_hot_targets(70 + i, S, C) for i in range(N)]))
    zl = (0.35 * torch.randn((N, C, crop, crop), generator=torch.Generator().manual_seed(5))).requires_grad_(True)
    lt = []
    for _ in range(loss_steps):
        zl.grad = None
        t0 = time.perf_counter()
        ce, mc = port.merged_positive_ce(zl, tg, sp, mk, 0.1, 'decomp')
        gr = port.group_max_ce(zl, tg, sp, mk, S, 0.1, 'onlymulti')
        (16.0 * ce + 8.0 * mc + gr).backward()
        lt.append(time.perf_counter() - t0)

    # model forward of one pool batch (same architecture in plain PyTorch ops on the host; parity pinned by G4)
    from mulactseg_amd.models import get_model
    net = get_model('deeplabv3pluswn_resnet50deepstem', C, 16, True, pretrained_backbone=False).eval()
    x = torch.randn((B, 3, H, W), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        t0 = time.perf_counter()
        net(x)
        t_model = time.perf_counter() - t0

    per_img = best / n_img
    return {"value": n_img * S / best, "unit": "superpixels/s", "cores": cores if t_all <= t_20 else min(20, cores), "kind": "port",
            "sample": "%d pictures = %d of the GPU leg's resident [%d,%d,%d,%d] batches copied to the host; scorer round = both passes + ban + "
                      "tuple list + Python sort + fair-counting budget walk (oracle/port.py, torch %s CPU); median of %d at %d threads; "
                      "+ %d loss steps and one model forward at the same thread count"
                      % (n_img, n_b, B, C, H, W, torch.__version__, reps, min(20, cores), loss_steps),
            "scorer_seconds": {"threads_%d" % min(20, cores): t_20, "threads_%d_scaled_from_%d_pictures" % (cores, n_all): t_all,
                               "runs": {"20": ts_20, "all": ts_all}},
            "extrapolated_pool_round_s": {"scorer_only": per_img * POOL_IMAGES,
                                          "with_two_model_forwards": (per_img + 2 * t_model / B) * POOL_IMAGES,
                                          "note": "linear in the picture count: %.3f s per picture x 2 975 (the reference runs the model "
                                                  "twice per picture, once per pass)" % per_img},
            "selection_vs_gpu": sel_cmp,
            "losses": {"seconds_per_step_fwd_bwd": float(np.median(lt)), "steps": loss_steps, "threads": lm_threads,
                       "shape": [N, C, crop, crop], "selected_fraction": float(mk.float().mean())},
            "model_forward": {"seconds_per_batch": t_model, "batch": [B, 3, H, W], "threads": lm_threads}}


def pmc_traffic(kernel, default_shape):
    """(HBM bytes per launch, source file) of `kernel` from the latest committed rocprofv3 --pmc summary
    (profiles/r*/..pmc_traffic.json, produced by profiles/summarize.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this same command, gfx950 FETCH_SIZE x2 correction applied).  (None, reason) when the bench shape is not the
    profiled default shape, no summary is present, or the summary was measured on OTHER kernel sources than the ones in this tree
    (it records the sha256 of the scan sources; a stale file is refused): PMC counters cannot be read from inside this process."""
    if not default_shape:
        return None, None
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_traffic.json")))
    for f in reversed(files):
        try:
            doc = json.load(open(f))
            k = doc["kernels"].get(kernel)
            if not k:
                continue
            shas = doc.get("kernel_source_sha16")
            if not shas:
                return None, "refused %s: it does not say which kernel sources it measured" % os.path.relpath(f, ROOT)
            for rel, sha in shas.items():
                if hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16] != sha:
                    return None, "refused %s: %s has changed since it was measured (re-run tools/profile_r05.sh scan)" % (os.path.relpath(f, ROOT), rel)
            return k["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def rank_launch_command(n_gpus, argv, port):
    """The command `python bench.py --gpus N` runs when no launcher is around it: one rank per GPU of this node over RCCL,
    rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """Parent of an N-rank run.  It never touches the GPU (the ranks are child processes; nothing is exec'ed from a process that
    has initialised HIP) and relays rank 0's JSON line through the inherited stdout; its exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(rank_launch_command(args.gpus, argv, port), env=env)


def main():
    args = parse()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to report a mislabelled run\n"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
        sys.exit(2)
    # stdout carries exactly ONE line, the JSON: library banners (RCCL prints its version block to stdout when the first
    # communicator is built) go to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    from mulactseg_amd.active_selection.engine import HipBackend
    backend = HipBackend(dev)
    B, C, H, W, S = args.batch, args.classes, args.height, args.width, args.nseg
    invT = ops.inv_temperature(0.1)
    bufs = [make_batch(1000 * rank + 17 * i + 1, B, C, H, W, S, args.id_dtype, dev) for i in range(args.nbuf)]

    p0 = torch.zeros((B, C), dtype=torch.int64, device=dev)
    c0 = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
    h0 = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
    def ramp():
        for k in range(args.ramp):
            z, spx = bufs[k % args.nbuf]
            ops.single_pass_accum(z, spx, S, invT, prob_sum=p0, class_sum=c0, hist=h0)
    timed, n_selected, dt = timed_scan_round(args, dev, rank, world, backend, bufs, ramp)

    sp_ms = float(np.mean([a.elapsed_time(b) for a, b in timed.events]))
    tail_ms = float(timed.tail_ev[0].elapsed_time(timed.tail_ev[1]))
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
    cls_w = torch.linspace(0.3, 1.0, C, device=dev)
    p2 = torch.zeros((B, C), dtype=torch.int64, device=dev)
    s2 = torch.zeros((B, S), dtype=torch.int64, device=dev)
    h2 = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
    k2_ms = time_kernel(lambda k=0: ops.class_prob_sum(bufs[k % args.nbuf][0], invT, out=p2))
    k3_ms = time_kernel(lambda k=0: ops.bvsb_region_accum(bufs[k % args.nbuf][0], bufs[k % args.nbuf][1], cls_w, S, invT,
                                                          score_sum=s2, hist=h2))
    k2_bytes = B * (C * H * W * 4)
    k3_bytes = B * (C * H * W * 4 + H * W * id_bytes + S * (8 + 4 * C))
    default_shape = (B, C, H, W, S, args.id_dtype) == (4, 20, 1024, 2048, 2048, "int64")
    traffic, traffic_src = pmc_traffic("k_single_pass", default_shape)
    traffic3, traffic3_src = pmc_traffic("k_bvsb_region_accum", default_shape)
    out = {
        "metric": "superpixels scored/sec", "value": args.steps * B * S * world / dt, "unit": "superpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "acquisition-round: PixBal+ban-ignore scorer on resident logits, Cityscapes pool shape [%d,%d,%d,%d] per step; "
                               "per rank K single-pass scans (class prior + region sums + histograms) of its shard, then exchange 1 "
                               "(all-gather class sums) + device class weights (k_class_weight) + weighted finalize + ban + exchange 2 (all-gather scores) + "
                               "replicated K4 (keys, radix sort, fair-counting budget walk) over all N*K*%d*%d regions"
                               % (B, C, H, W, B, S),
                   "images_per_step": B, "logits": [B, C, H, W], "nseg": S, "id_dtype": args.id_dtype,
                   "temperature": 0.1, "sharding": "pool images across ranks, whole reference batches (engine.ShardPlan)",
                   "regions_scored": world * args.steps * B * S, "budget_clicks": timed.budget, "regions_selected": n_selected,
                   "round_tail_ms": tail_ms},
        "roofline": {"bound": "hbm", "kernel": "k_single_pass", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": sp_bytes, "avg_launch_ms": sp_ms},
        "two_pass": {"k_class_prob_sum": {"avg_launch_ms": k2_ms, "achieved_GBs": k2_bytes / (k2_ms * 1e-3) / 1e9,
                                          "bytes_per_launch": k2_bytes},
                     "k_bvsb_region_accum": {"avg_launch_ms": k3_ms, "achieved_GBs": k3_bytes / (k3_ms * 1e-3) / 1e9,
                                             "frac_of_peak": k3_bytes / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "bytes_per_launch": k3_bytes, "traffic": traffic3, "traffic_source": traffic3_src},
                     "superpixels_per_s": B * S * world / ((k2_ms + k3_ms) * 1e-3)},
    }

    def secondary(fn, *a):
        """Secondary legs never take the primary line down with them."""
        try:
            return fn(*a)
        except Exception as e:          # noqa: BLE001
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    cpu_bufs = bufs if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    if cpu_bufs is None:
        del bufs
    del timed
    torch.cuda.empty_cache()
    if not args.no_pool:
        out["pool_round"] = {"metric": "acquisition round over the fixed %d-picture x %d-superpixel pool, %d clicks, through "
                                       "RegionSelector.select_next_batch; sharded over %d rank(s)" % (POOL_IMAGES, S, POOL_CLICKS, world),
                             "scaling": "strong",
                             # the first round of a process pays first-use costs that no later round of the loop sees (code objects
                             # of the ordering / walk kernels and of the ATen ops behind the tail, allocator growth: ~0.05 s): one
                             # round over a 64-picture pool of the same shapes is run before the timed one, as the W warm-up steps are
                             "warm_up": "one untimed round over a %d-picture pool (%d clicks) before the timed ones" % (POOL_WARM_IMAGES, POOL_WARM_CLICKS),
                             "scan_only": (secondary(pool_round_bench, args, dev, rank, world, False, POOL_WARM_IMAGES, POOL_WARM_CLICKS),
                                           secondary(pool_round_bench, args, dev, rank, world, False))[1],
                             "with_model_forward": None if args.no_train else secondary(pool_round_bench, args, dev, rank, world, True)}
        torch.cuda.empty_cache()
    def leg(fn, *a):
        # every leg starts from an empty caching allocator: the blocks a previous leg left behind fit the next leg's shapes badly (the
        # 769 crop after the 768 one) and what the allocator then carves out of them is scattered -- not part of any timed region
        gc.collect()
        torch.cuda.empty_cache()
        return secondary(fn, *a)
    out["train_iter"] = None if (args.no_train or args.no_trainleg) else leg(train_iter_bench, args, dev, world, args.crop)
    out["train_iter_769"] = None if (args.no_train or args.no_trainleg) else leg(train_iter_bench, args, dev, world, 769)
    out["acquisition_with_model"] = None if args.no_train else leg(acquisition_with_model_bench, args, dev, world)
    out["stage2"] = None if (args.no_train or args.no_trainleg or rank != 0) else secondary(stage2_bench, args, dev)
    # the secondary legs' headline numbers as TOP-LEVEL keys (a reader of the parsed line need not dig through the nested objects)
    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    out["train_iter_ms_768"] = pick(out, "train_iter", "ms_per_iter")
    out["train_iter_images_per_s_768"] = pick(out, "train_iter", "value")
    out["train_iter_mfma_frac_768"] = pick(out, "train_iter", "mfma", "mfma_frac")
    out["train_iter_ms_769"] = pick(out, "train_iter_769", "ms_per_iter")
    out["train_iter_images_per_s_769"] = pick(out, "train_iter_769", "value")
    out["pool_forward_ms_per_batch"] = pick(out, "acquisition_with_model", "ms_per_batch")
    out["pool_forward_mfma_frac"] = pick(out, "acquisition_with_model", "mfma", "mfma_frac")
    out["pool_round_scan_only_s"] = pick(out, "pool_round", "scan_only", "seconds")
    out["pool_round_with_model_s"] = pick(out, "pool_round", "with_model_forward", "seconds")
    out["loss_gpu_ms_fwd_bwd"] = pick(out, "train_iter", "loss_only", "gpu_ms_fwd_bwd")
    if rank == 0:
        out["cpu_baseline"] = secondary(cpu_baseline, args, dev, cpu_bufs, backend) if cpu_bufs is not None else None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    d = _dist()
    if d is not None:
        if world > 1:
            d.barrier()
        d.destroy_process_group()


if __name__ == "__main__":
    main()
