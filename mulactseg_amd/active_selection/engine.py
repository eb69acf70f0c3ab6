"""Acquisition-round engine: the device-side state and the multi-GPU sharding behind the
``RegionSelector`` plugins.

One process per GPU.  The unlabeled pool is sharded over ranks in whole *reference batches*
(``val_batch_size`` consecutive pool images, ``active_selection/utils.py:47-57``) so that the
reference's "mean of per-batch means" class prior (``my_bvsb_predclsbal_pwr_banignore.py:42-45``)
and the (image rank, id) tie-break keys are identical for any number of GPUs.  The scan itself needs
no collective; the round has exactly two tiny exchanges:

  1. after pass 1: all-gather of the per-image fixed-point class sums  [N_img, C] int64  (476 KB for
     the Cityscapes pool) -> every rank derives the same class weights in f64 (``k_class_weight``, on the device);
  2. after pass 2: all-gather of the per-image region scores [N_img, S] f32 (24 MB) -> every rank
     runs the same K4 ordering + budget walk (replicated, deterministic).

Both are latency-bound RCCL collectives over xGMI (``torch.distributed`` backend "nccl").

The compute backend is an explicit object.  The only backend in this package is ``HipBackend``
(the C ABI of ``libmulactseg_hip.so``); it raises if the library or a GPU is missing -- there is no
CPU fallback.  (The CPU test-suite injects an oracle-backed stand-in to exercise the sharding and
merge logic under ``gloo``.)
"""
import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# sharding
# ------------------------------------------------------------------------------------------------
class ShardPlan:
    """Contiguous blocks of whole reference batches per rank."""

    def __init__(self, n_img, batch_size, rank=0, world=1):
        self.n_img, self.batch_size, self.rank, self.world = n_img, batch_size, rank, world
        self.n_batches = (n_img + batch_size - 1) // batch_size
        per = (self.n_batches + world - 1) // world
        self.batch_lo = min(rank * per, self.n_batches)
        self.batch_hi = min((rank + 1) * per, self.n_batches)
        self.img_lo = min(self.batch_lo * batch_size, n_img)
        self.img_hi = min(self.batch_hi * batch_size, n_img)
        self.per_rank_imgs = per * batch_size                      # padded shard length for gathers
        self.batch_of = (np.arange(n_img) // batch_size).astype(np.int32)

    @property
    def local_indices(self):
        return list(range(self.img_lo, self.img_hi))

    @property
    def n_local(self):
        return self.img_hi - self.img_lo


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def current_rank_world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def gather_rows(local, plan):
    """All-gather row blocks ``local`` [n_local, ...] into [n_img, ...] (identical on every rank)."""
    d = _dist()
    if d is None or plan.world == 1:
        return local
    pad = torch.zeros((plan.per_rank_imgs,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:plan.n_local] = local
    out = torch.empty((plan.world * plan.per_rank_imgs,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    d.all_gather_into_tensor(out, pad)
    return out[:plan.n_img].contiguous()


# ------------------------------------------------------------------------------------------------
# the class-weight arithmetic between the passes, restated for the host (tests, CPU stand-in backends)
# ------------------------------------------------------------------------------------------------
def class_weight_from_sums(prob_sum, hw, batch_of, n_batches, coeff):
    """Reference: ``cum = sum_b mean_b / len(loader)``; ``cls_weight = (coeff*cum + 1)**-2``
    (``my_bvsb_predclsbal_pwr_banignore.py:42,45,47``), evaluated in f64 from the integer per-image
    sums (23 fractional bits) and rounded once to f32.  Same operation order as
    ``oracle/exact.c:exact_class_weight`` (bit-identical)."""
    prob_sum = np.ascontiguousarray(prob_sum).view(np.uint64)
    n_img, C = prob_sum.shape
    batch_of = np.asarray(batch_of)
    # exact integer sums per reference batch, then the batch means added IN BATCH ORDER (np.cumsum is strictly
    # sequential; np.sum is pairwise and would round differently from the oracle's loop)
    sums = np.zeros((n_batches, C), dtype=np.uint64)
    np.add.at(sums, batch_of, prob_sum)
    n_in = np.bincount(batch_of, minlength=n_batches)[:n_batches]
    live = n_in > 0
    terms = (sums[live].astype(np.float64) / np.float64(8388608.0)) / (n_in[live].astype(np.float64) * np.float64(hw))[:, None]
    acc = np.cumsum(terms, axis=0)[-1] if terms.shape[0] else np.zeros(C, dtype=np.float64)
    cum = acc / np.float64(n_batches)
    t = np.float64(coeff) * cum + np.float64(1.0)
    return cum, (np.float64(1.0) / (t * t)).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# backend
# ------------------------------------------------------------------------------------------------
class HipBackend:
    """The C ABI of libmulactseg_hip.so on one GPU (see include/mulactseg_hip.h)."""
    name = "hip"

    def __init__(self, device):
        from .. import _lib, ops
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.MulActSegHipError("HipBackend needs a ROCm device, got %s (no CPU path exists)" % device)
        _lib.load()
        self.ops = ops

    def inv_temperature(self, T):
        return self.ops.inv_temperature(T)

    def class_prob_sum(self, logits, invT, out):
        self.ops.class_prob_sum(logits, invT, out=out)

    def region_accum(self, logits, spx, cls_w, S, invT, score_sum, hist):
        self.ops.bvsb_region_accum(logits, spx, cls_w, S, invT, score_sum=score_sum, hist=hist)

    def finalize(self, score_sum, hist, ban_class, want_hist_i64=False):
        return self.ops.region_finalize(score_sum, hist, ban_class, want_hist_i64)

    def single_pass(self, logits, spx, S, invT, prob_sum, class_sum, hist):
        self.ops.single_pass_accum(logits, spx, S, invT, prob_sum=prob_sum, class_sum=class_sum, hist=hist)

    def single_pass_lowres(self, zq, size, spx, S, invT, prob_sum, class_sum, hist):
        self.ops.single_pass_accum_lowres(zq, size, spx, S, invT, prob_sum=prob_sum, class_sum=class_sum, hist=hist)

    def class_weight(self, prob_sum, hw, batch_size, n_batches, coeff):
        """(cum f64 [C], cls_w f32 [C]) on the device from the gathered per-picture class sums; the integer form of the
        weights rides along on the tensor so that ``finalize_weighted`` needs no conversion (and no host round trip)."""
        cum, w, w31 = self.ops.class_weight(prob_sum.contiguous(), hw, batch_size, n_batches, coeff)
        w._mas_w31 = w31
        return cum, w

    def finalize_weighted(self, class_sum, hist, cls_w, ban_class, want_hist_i64=False):
        w31 = None if cls_w is None else getattr(cls_w, '_mas_w31', None)
        if w31 is None:     # weights that did not come from class_weight(): floor(w * 2^31) with exact device arithmetic
            w = torch.ones(hist.shape[-1], dtype=torch.float32, device=hist.device) if cls_w is None else cls_w.detach().to(hist.device)
            w31 = (w.double() * 2147483648.0).floor().to(torch.int64).view(torch.int32)[::2].contiguous()
        return self.ops.region_finalize_weighted(class_sum, hist, w31, ban_class, want_hist_i64)

    def minmax_normalize_(self, scores):
        return self.ops.minmax_normalize_(scores)

    def region_reweight_(self, scores, dominant, ban_class, cls_w):
        return self.ops.region_reweight_(scores, dominant, ban_class, cls_w)

    def dominant_hist(self, dominant, C):
        return self.ops.dominant_hist(dominant, C)

    def select(self, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out):
        keys = self.ops.region_keys(scores, valid, img_rank)
        skeys = self.ops.sort_keys_desc(keys)
        nsel, simg, sid, ssc = self.ops.budget_walk(skeys, region_cost, img_of_rank, scores.shape[1], budget, max_out)
        n = int(nsel.item())               # the round's single host synchronisation
        return n, simg[:n].cpu().numpy(), sid[:n].cpu().numpy(), ssc[:n].cpu().numpy()

    def local_head(self, plan, scores, valid, img_rank, max_out):
        """int64 [max_out]: the first ``max_out`` keys of the descending order of THIS rank's rows (zero-padded: key 0 = "no
        region", sorts last and costs nothing)."""
        lo, hi = plan.img_lo, plan.img_hi
        head = torch.zeros(max_out, dtype=torch.int64, device=scores.device)
        if hi > lo:
            keys = self.ops.region_keys(scores[lo:hi].contiguous(), None if valid is None else valid[lo:hi].contiguous(),
                                        img_rank[lo:hi].contiguous())
            m = min(keys.numel(), max_out)
            head[:m] = self.ops.sort_keys_desc(keys)[:m]
        return head

    def walk_heads(self, heads, region_cost, img_of_rank, S, budget, max_out):
        """Merged ordering + budget walk over the concatenated per-rank heads."""
        nsel, simg, sid, ssc = self.ops.budget_walk(self.ops.sort_keys_desc(heads), region_cost, img_of_rank, S, budget, max_out)
        n = int(nsel.item())               # the round's single host synchronisation
        return n, simg[:n].cpu().numpy(), sid[:n].cpu().numpy(), ssc[:n].cpu().numpy()

    def select_sharded(self, plan, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out):
        """See ``select_regions``: local keys + local sort, all-gather of the per-rank heads (RCCL), merged sort + walk."""
        head = self.local_head(plan, scores, valid, img_rank, max_out)
        merged = torch.empty(plan.world * max_out, dtype=torch.int64, device=scores.device)
        _dist().all_gather_into_tensor(merged, head)
        return self.walk_heads(merged, region_cost, img_of_rank, scores.shape[1], budget, max_out)


def default_backend(device):
    return HipBackend(device)


def select_regions(backend, plan, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out):
    """K4 over the whole pool: the consumed prefix (n, picture indices, ids, scores) of the descending (score, path rank, id)
    order under the click budget.  One rank: ``backend.select``.  Several ranks: every rank orders only ITS rows, the first
    ``max_out`` keys of every rank are all-gathered and the merged candidates are ordered and walked on every rank -- the
    prefix the budget consumes has at most ``max_out`` regions, so it lies inside the union of the per-rank heads and the
    result equals the replicated ordering of all regions, at 1 / world of the sorting work.  (``max_out`` None -- a region may
    cost nothing -- falls back to the replicated ordering.)"""
    if plan.world > 1 and max_out is not None and _dist() is not None and hasattr(backend, 'select_sharded'):
        return backend.select_sharded(plan, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out)
    return backend.select(scores, valid, img_rank, img_of_rank, region_cost, budget, max_out)


# ------------------------------------------------------------------------------------------------
# one acquisition round
# ------------------------------------------------------------------------------------------------
class AcquisitionRound:
    """Per-round accumulators for the images of this rank plus the two exchanges.

    Usage (what the selector plugins do):
        rnd = AcquisitionRound(n_img, C, S, batch_size, temperature, backend)
        for k, logits in pass 1:  rnd.add_prior(first_local_row, logits)
        w = rnd.class_weights(coeff)                 # exchange 1
        for k, logits in pass 2:  rnd.add_regions(first_local_row, logits, spx, w)
        scores = rnd.scores(ban_class)               # exchange 2 -> [n_img, S] on every rank
    """

    def __init__(self, n_img, n_channels, n_superpixels, batch_size, temperature, backend, rank=None, world=None,
                 single_pass=False):
        r, w = current_rank_world()
        self.plan = ShardPlan(n_img, batch_size, r if rank is None else rank, w if world is None else world)
        self.C, self.S = n_channels, n_superpixels
        self.backend = backend
        self.invT = backend.inv_temperature(temperature)
        self.single_pass = single_pass
        dev = backend.device
        nl = max(self.plan.n_local, 1)
        self.prob_sum = torch.zeros((nl, n_channels), dtype=torch.int64, device=dev)
        self.hist = torch.zeros((nl, n_superpixels, n_channels), dtype=torch.int32, device=dev)
        if single_pass:     # per (region, arg-max class) sums of the unweighted margin: 327 KB per Cityscapes image
            self.class_sum = torch.zeros((nl, n_superpixels, n_channels), dtype=torch.int64, device=dev)
        else:
            self.score_sum = torch.zeros((nl, n_superpixels), dtype=torch.int64, device=dev)
        self.hw = None

    def _rows(self, row0, logits):
        B = logits.shape[0]
        if row0 < 0 or row0 + B > self.plan.n_local:
            raise IndexError("batch rows [%d,%d) outside this rank's shard of %d images" % (row0, row0 + B, self.plan.n_local))
        hw = logits.shape[2] * logits.shape[3]
        if self.hw is None:
            self.hw = hw
        elif self.hw != hw:
            raise ValueError("all pool images of one round must share H*W (the class prior is a pixel mean)")
        return slice(row0, row0 + B)

    def add_prior(self, row0, logits):
        r = self._rows(row0, logits)
        self.backend.class_prob_sum(logits.contiguous(), self.invT, self.prob_sum[r])

    def class_weights(self, coeff):
        allsum = gather_rows(self.prob_sum[:self.plan.n_local], self.plan)        # exchange 1
        hw = self.hw
        d = _dist()
        if d is not None and self.plan.world > 1:
            t = torch.tensor([hw or 0], dtype=torch.int64, device=allsum.device)
            d.all_reduce(t, op=d.ReduceOp.MAX)
            hw = int(t.item())
        self._cum, w = self.backend.class_weight(allsum, hw, self.plan.batch_size, self.plan.n_batches, coeff)
        return w

    @property
    def cum(self):
        """f64 [C] numpy: the class prior ``cumulated_pred_prob / len(loader)`` of the last ``class_weights`` call."""
        c = self._cum
        return c.cpu().numpy() if torch.is_tensor(c) else c

    def add_regions(self, row0, logits, spx, cls_w):
        r = self._rows(row0, logits)
        self.backend.region_accum(logits.contiguous(), spx.contiguous(), cls_w, self.S, self.invT, self.score_sum[r], self.hist[r])

    def add_single_pass(self, row0, logits, spx):
        """One scan per batch: class-probability sums + per (region, class) margin sums + histogram."""
        r = self._rows(row0, logits)
        self.backend.single_pass(logits.contiguous(), spx.contiguous(), self.S, self.invT, self.prob_sum[r],
                                 self.class_sum[r], self.hist[r])

    def add_single_pass_lowres(self, row0, zq, size, spx):
        """As ``add_single_pass`` for the logits ``F.interpolate(zq, size, 'bilinear', align_corners=False)`` (the model's final
        upsampling, ``models/segmentation/utils.py:25``) evaluated inside the scan: same accumulators, bit for bit."""
        B = zq.shape[0]
        if row0 < 0 or row0 + B > self.plan.n_local:
            raise IndexError("batch rows [%d,%d) outside this rank's shard of %d images" % (row0, row0 + B, self.plan.n_local))
        hw = int(size[0]) * int(size[1])
        if self.hw is None:
            self.hw = hw
        elif self.hw != hw:
            raise ValueError("all pool images of one round must share H*W (the class prior is a pixel mean)")
        r = slice(row0, row0 + B)
        self.backend.single_pass_lowres(zq.contiguous(), (int(size[0]), int(size[1])), spx.contiguous(), self.S, self.invT,
                                        self.prob_sum[r], self.class_sum[r], self.hist[r])

    def scores_single_pass(self, cls_w, ban_class=-1, want_hist=False):
        """Weighted region means from the single-pass accumulators (cls_w None -> unweighted)."""
        n = self.plan.n_local
        if n == 0:
            dev = self.backend.device
            score = torch.zeros((0, self.S), dtype=torch.float32, device=dev)
            h64 = torch.zeros((0, self.S, self.C), dtype=torch.int64, device=dev)
        else:
            score, dom, cnt, h64 = self.backend.finalize_weighted(self.class_sum[:n], self.hist[:n], cls_w, ban_class, want_hist)
        full = gather_rows(score, self.plan)                                      # exchange 2
        if want_hist:
            return full, gather_rows(h64, self.plan)
        return full

    def scores(self, ban_class=-1, want_hist=False, want_dominant=False):
        n = self.plan.n_local
        if n == 0:      # more ranks than reference batches: this rank only takes part in the exchanges
            dev = self.backend.device
            score = torch.zeros((0, self.S), dtype=torch.float32, device=dev)
            dom = torch.zeros((0, self.S), dtype=torch.int32, device=dev)
            h64 = torch.zeros((0, self.S, self.C), dtype=torch.int64, device=dev)
        else:
            score, dom, cnt, h64 = self.backend.finalize(self.score_sum[:n], self.hist[:n], ban_class, want_hist)
        full = gather_rows(score, self.plan)                                      # exchange 2
        if want_dominant:
            return full, gather_rows(dom, self.plan)
        if want_hist:
            return full, gather_rows(h64, self.plan)
        return full
