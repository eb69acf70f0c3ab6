"""Tensor-level wrappers over the C ABI: contiguous ROCm tensors in, ROCm tensors out.

Every function launches on the current torch stream and never synchronises with the host.
These are the only callers of ``_lib``; the plugin-level code (``active_selection``, ``utils.loss``)
is written against this module.
"""
import os

import numpy as np
import torch

from . import _lib

_ID_CODES = {torch.int64: _lib.ID_I64, torch.int32: _lib.ID_I32, torch.uint16: _lib.ID_U16,
             torch.int16: _lib.ID_U16}


def inv_temperature(T):
    """invT as the ABI defines it: float32(1 / float32(T))."""
    return float(np.float32(1.0) / np.float32(T))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(t, name, dtype=None):
    if not t.is_cuda:
        raise _lib.MulActSegHipError("%s must live on the GPU (no CPU path exists)" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _id_code(spx):
    try:
        return _ID_CODES[spx.dtype]
    except KeyError:
        raise TypeError("superpixel ids must be int64, int32 or uint16, got %s" % spx.dtype)


def class_prob_sum(z, invT, out=None):
    """K2: per-image fixed-point class-probability sums ``[B,C]`` (int64 holding uint64 bits).
    Reference: my_bvsb_predclsbal_pwr_banignore.py:41-42."""
    _need(z, "z", torch.float32)
    B, C, H, W = z.shape
    if out is None:
        out = torch.zeros((B, C), dtype=torch.int64, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().mas_class_prob_sum(z.data_ptr(), B, C, H, W, invT, out.data_ptr(), _stream(z)),
                   "mas_class_prob_sum")
    return out


def bvsb_region_accum(z, spx, cls_w, S, invT, score_sum=None, hist=None):
    """K1+K3: fixed-point region sums ``[B,S]`` (int64 bits of uint64) and arg-max-class histogram
    ``[B,S,C]`` (int32 bits of uint32).  Reference: my_bvsb.py:19-27, ..._pwr_banignore.py:57-69."""
    _need(z, "z", torch.float32)
    _need(spx, "spx")
    B, C, H, W = z.shape
    if tuple(spx.shape) != (B, H, W):
        raise ValueError("spx shape %s does not match logits %s" % (tuple(spx.shape), tuple(z.shape)))
    if cls_w is not None:
        _need(cls_w, "cls_w", torch.float32)
        if cls_w.numel() != C:
            raise ValueError("cls_w must have C=%d entries" % C)
    if score_sum is None:
        score_sum = torch.zeros((B, S), dtype=torch.int64, device=z.device)
    if hist is None:
        hist = torch.zeros((B, S, C), dtype=torch.int32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().mas_bvsb_region_accum(
            z.data_ptr(), spx.data_ptr(), _id_code(spx), cls_w.data_ptr() if cls_w is not None else None,
            B, C, H, W, S, invT, score_sum.data_ptr(), hist.data_ptr(), _stream(z)), "mas_bvsb_region_accum")
    return score_sum, hist


def region_finalize(score_sum, hist, ban_class=-1, want_hist_i64=False):
    """K3 tail + ban: returns (score f32, dominant i32, count i32, hist_i64 or None), shaped like
    ``score_sum``.  Reference: ..._pwr_banignore.py:79-84."""
    _need(score_sum, "score_sum", torch.int64)
    _need(hist, "hist", torch.int32)
    C = hist.shape[-1]
    n = score_sum.numel()
    dev = score_sum.device
    score = torch.empty(score_sum.shape, dtype=torch.float32, device=dev)
    dom = torch.empty(score_sum.shape, dtype=torch.int32, device=dev)
    cnt = torch.empty(score_sum.shape, dtype=torch.int32, device=dev)
    h64 = torch.empty(hist.shape, dtype=torch.int64, device=dev) if want_hist_i64 else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mas_region_finalize(
            score_sum.data_ptr(), hist.data_ptr(), n, C, ban_class, score.data_ptr(), dom.data_ptr(),
            cnt.data_ptr(), h64.data_ptr() if h64 is not None else None, _stream(score_sum)),
            "mas_region_finalize")
    return score, dom, cnt, h64


# ------------------------------------------------------------------------------------------------
# stage-1 partial-label losses
# ------------------------------------------------------------------------------------------------
def target_bits(targets, cols_used=None):
    """u8 multi-hot rows [..., cols] -> int32 bit masks [...] over the first ``cols_used`` columns."""
    _need(targets, "targets", torch.uint8)
    cols = targets.shape[-1]
    cols_used = cols if cols_used is None else cols_used
    bits = torch.empty(targets.shape[:-1], dtype=torch.int32, device=targets.device)
    with torch.cuda.device(targets.device):
        _lib.check(_lib.load().mas_target_bits(targets.data_ptr(), bits.numel(), cols, cols_used, bits.data_ptr(),
                                               _stream(targets)), "mas_target_bits")
    return bits


def _mask_u8(mask):
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    return _need(mask, "spmasks", torch.uint8)


def partial_loss_fwd(z, spx, mask, bits, invT, flags, reduce_acc=None):
    """Forward scan + group finalize + loss values.  Returns (losses f32[3], acc i64[8], gmax i64[N,S,C])
    -- all on the device, no host synchronisation.  ``reduce_acc(acc)`` (optional) runs between the scans
    and the division: data-parallel training all-reduces the integer sums / counts there so that the
    normalisers 1 + n are global over the batch, as on one GPU."""
    _need(z, "inputs", torch.float32)
    _need(spx, "superpixels")
    mask = _mask_u8(mask)
    _need(bits, "bits", torch.int32)
    N, C, H, W = z.shape
    S = bits.shape[1]
    if tuple(spx.shape) != (N, H, W) or tuple(mask.shape) != (N, H, W) or bits.shape[0] != N:
        raise ValueError("shape mismatch between inputs %s, superpixels %s, spmasks %s, targets %s"
                         % (tuple(z.shape), tuple(spx.shape), tuple(mask.shape), tuple(bits.shape)))
    dev = z.device
    acc = torch.zeros(_lib.ACC_WORDS, dtype=torch.int64, device=dev)
    gmax = torch.zeros((N, S, C), dtype=torch.int64, device=dev) if flags & _lib.LOSS_GROUP else None
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        st = _stream(z)
        _lib.check(lib.mas_partial_loss_fwd(z.data_ptr(), spx.data_ptr(), _id_code(spx), mask.data_ptr(), bits.data_ptr(),
                                            N, C, H, W, S, invT, flags, gmax.data_ptr() if gmax is not None else None,
                                            acc.data_ptr(), st), "mas_partial_loss_fwd")
        if gmax is not None:
            _lib.check(lib.mas_group_finalize(gmax.data_ptr(), gmax.numel(), acc.data_ptr(), st), "mas_group_finalize")
        if reduce_acc is not None:
            reduce_acc(acc)
        _lib.check(lib.mas_loss_values(acc.data_ptr(), flags, losses.data_ptr(), st), "mas_loss_values")
    return losses, acc, gmax


def partial_loss_bwd(z, spx, mask, bits, gmax, acc, grad_out, invT, flags):
    """Backward scan: dz [N,C,H,W] for upstream gradients ``grad_out`` f32[3] (device tensor)."""
    mask = _mask_u8(mask)
    _need(grad_out, "grad_out", torch.float32)
    N, C, H, W = z.shape
    S = bits.shape[1]
    dev = z.device
    scale = torch.empty(3, dtype=torch.float32, device=dev)
    dz = torch.empty_like(z)
    lib = _lib.load()
    with torch.cuda.device(dev):
        st = _stream(z)
        _lib.check(lib.mas_loss_scales(acc.data_ptr(), grad_out.data_ptr(), flags, scale.data_ptr(), st), "mas_loss_scales")
        _lib.check(lib.mas_partial_loss_bwd(z.data_ptr(), spx.data_ptr(), _id_code(spx), mask.data_ptr(), bits.data_ptr(),
                                            gmax.data_ptr() if gmax is not None else None, scale.data_ptr(),
                                            N, C, H, W, S, invT, flags, dz.data_ptr(), st), "mas_partial_loss_bwd")
    return dz


class LossState:
    """What a fused forward leaves for its backward: the work buffer (accumulators | arg-pixel table | target bit masks) and views."""
    __slots__ = ("work", "acc", "gmax", "bits", "N", "S", "C", "flags")


def partial_loss_fwd_fused(z, size, spx, mask, invT, flags, targets=None, cols_used=None, bits=None, weights=None, reduce_acc=None):
    """Everything the forward direction launches in ONE library call (mas_partial_loss_fwd_fused: prep, scan, finalize + values).
    ``z`` [N,C,H,W] with ``size`` None, or the quarter-resolution logits [N,C,h,w] with ``size`` = (H, W).  Either ``targets``
    (u8 [N,S,cols], masks over the first ``cols_used`` columns are formed in the prep launch) or ready ``bits`` (int32 [N,S]).
    ``weights`` f32 [3]: losses gets a fourth entry, the weighted objective.  ``reduce_acc`` (data parallel): called on the
    accumulators between the scans and the division.  Returns (losses, LossState)."""
    _need(z, "inputs", torch.float32)
    _need(spx, "superpixels")
    mask = _mask_u8(mask)
    N, C = z.shape[0], z.shape[1]
    low = size is not None
    H, W = (int(size[0]), int(size[1])) if low else (z.shape[2], z.shape[3])
    h, w = (z.shape[2], z.shape[3]) if low else (0, 0)
    if targets is not None:
        _need(targets, "targets", torch.uint8)
        S, cols = targets.shape[1], targets.shape[2]
        cols_used = cols if cols_used is None else int(cols_used)
        if targets.shape[0] != N:
            raise ValueError("targets %s do not match the batch of %d" % (tuple(targets.shape), N))
    else:
        _need(bits, "bits", torch.int32)
        S, cols, cols_used = bits.shape[1], 0, 0
        if bits.shape[0] != N:
            raise ValueError("bits %s do not match the batch of %d" % (tuple(bits.shape), N))
    if tuple(spx.shape) != (N, H, W) or tuple(mask.shape) != (N, H, W):
        raise ValueError("shape mismatch between inputs %s at size %s, superpixels %s, spmasks %s"
                         % (tuple(z.shape), (H, W), tuple(spx.shape), tuple(mask.shape)))
    dev = z.device
    lib = _lib.load()
    nbytes = int(lib.mas_partial_loss_work_bytes(N, S, C, flags))
    work = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)          # (zeroed by the prep launch)
    losses = torch.empty(3 if weights is None else 4, dtype=torch.float32, device=dev)
    if weights is not None:
        _need(weights, "weights", torch.float32)
    st8 = LossState()
    st8.work, st8.acc, st8.N, st8.S, st8.C, st8.flags = work, work[:_lib.ACC_WORDS], N, S, C, flags
    st8.gmax = work[_lib.ACC_WORDS:_lib.ACC_WORDS + N * S * C].view(N, S, C) if flags & _lib.LOSS_GROUP else None
    st8.bits = bits
    with torch.cuda.device(dev):
        st = _stream(z)
        _lib.check(lib.mas_partial_loss_fwd_fused(z.data_ptr(), h, w, spx.data_ptr(), _id_code(spx), mask.data_ptr(), _opt(targets), cols, cols_used,
                                                  _opt(bits), N, C, H, W, S, invT, flags, _opt(weights), work.data_ptr(), work.numel() * 8,
                                                  None if reduce_acc is not None else losses.data_ptr(), st), "mas_partial_loss_fwd_fused")
        if reduce_acc is not None:
            reduce_acc(st8.acc)
            if weights is None:
                _lib.check(lib.mas_loss_values(st8.acc.data_ptr(), flags, losses.data_ptr(), st), "mas_loss_values")
            else:
                _lib.check(lib.mas_loss_values_weighted(st8.acc.data_ptr(), flags, weights.data_ptr(), losses.data_ptr(), st), "mas_loss_values_weighted")
    return losses, st8


def partial_loss_bwd_fused(z, size, spx, mask, state, grad, invT, weights=None, want_fix=False):
    """The backward direction in ONE library call (mas_partial_loss_bwd_fused): dz [N,C,H,W] (``size`` None) or dzq [N,C,h,w]."""
    mask = _mask_u8(mask)
    _need(grad, "grad_out", torch.float32)
    N, C = z.shape[0], z.shape[1]
    low = size is not None
    H, W = (int(size[0]), int(size[1])) if low else (z.shape[2], z.shape[3])
    h, w = (z.shape[2], z.shape[3]) if low else (0, 0)
    dz = torch.empty_like(z)
    fix = torch.empty((N, C, h, w), dtype=torch.int64, device=z.device) if low else None      # (zeroed inside the call)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().mas_partial_loss_bwd_fused(z.data_ptr(), h, w, spx.data_ptr(), _id_code(spx), mask.data_ptr(), _opt(state.bits),
                                                          state.work.data_ptr(), grad.data_ptr(), _opt(weights), N, C, H, W, state.S, invT,
                                                          state.flags, dz.data_ptr(), _opt(fix), _stream(z)), "mas_partial_loss_bwd_fused")
    return (dz, fix) if want_fix else dz


# ------------------------------------------------------------------------------------------------
# K4: ordering + budgeted selection walk
# ------------------------------------------------------------------------------------------------
def partial_loss_fwd_lowres(zq, size, spx, mask, bits, invT, flags, reduce_acc=None, weights=None):
    """As ``partial_loss_fwd`` for the logits ``F.interpolate(zq, size, 'bilinear', align_corners=False)`` without
    materialising them: ``zq`` [N,C,h,w] quarter-resolution logits, ``size`` = (H, W) of ids / masks.  ``weights`` (f32 [3]
    device tensor: w_ce, w_mc, w_group): ``losses`` gets a fourth entry, the weighted objective (w_ce*ce + w_mc*mc) + w_group*group."""
    _need(zq, "inputs", torch.float32)
    _need(spx, "superpixels")
    mask = _mask_u8(mask)
    _need(bits, "bits", torch.int32)
    N, C, h, w = zq.shape
    H, W = int(size[0]), int(size[1])
    S = bits.shape[1]
    if tuple(spx.shape) != (N, H, W) or tuple(mask.shape) != (N, H, W) or bits.shape[0] != N:
        raise ValueError("shape mismatch between inputs %s at size %s, superpixels %s, spmasks %s, targets %s"
                         % (tuple(zq.shape), (H, W), tuple(spx.shape), tuple(mask.shape), tuple(bits.shape)))
    dev = zq.device
    if flags & _lib.LOSS_GROUP:             # accumulators and the arg-pixel table from ONE zero-filled allocation (one memset)
        buf = torch.zeros(_lib.ACC_WORDS + N * S * C, dtype=torch.int64, device=dev)
        acc, gmax = buf[:_lib.ACC_WORDS], buf[_lib.ACC_WORDS:].view(N, S, C)
    else:
        acc, gmax = torch.zeros(_lib.ACC_WORDS, dtype=torch.int64, device=dev), None
    losses = torch.empty(3 if weights is None else 4, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        st = _stream(zq)
        _lib.check(lib.mas_partial_loss_fwd_lowres(zq.data_ptr(), h, w, spx.data_ptr(), _id_code(spx), mask.data_ptr(), bits.data_ptr(),
                                                   N, C, H, W, S, invT, flags, gmax.data_ptr() if gmax is not None else None,
                                                   acc.data_ptr(), st), "mas_partial_loss_fwd_lowres")
        if gmax is not None:
            _lib.check(lib.mas_group_finalize(gmax.data_ptr(), gmax.numel(), acc.data_ptr(), st), "mas_group_finalize")
        if reduce_acc is not None:
            reduce_acc(acc)
        if weights is None:
            _lib.check(lib.mas_loss_values(acc.data_ptr(), flags, losses.data_ptr(), st), "mas_loss_values")
        else:
            _need(weights, "weights", torch.float32)
            _lib.check(lib.mas_loss_values_weighted(acc.data_ptr(), flags, weights.data_ptr(), losses.data_ptr(), st), "mas_loss_values_weighted")
    return losses, acc, gmax


def partial_loss_bwd_lowres(zq, size, spx, mask, bits, gmax, acc, grad_out, invT, flags, want_fix=False, weights=None):
    """Gradient of the losses with respect to the quarter-resolution logits: dzq [N,C,h,w] f32 (and the int64 fixed-point
    sums it was rounded from when ``want_fix``).  With ``weights`` [3], ``grad_out`` is the upstream gradient of the weighted
    objective, a one-element tensor."""
    mask = _mask_u8(mask)
    _need(grad_out, "grad_out", torch.float32)
    N, C, h, w = zq.shape
    H, W = int(size[0]), int(size[1])
    S = bits.shape[1]
    dev = zq.device
    scale = torch.empty(3, dtype=torch.float32, device=dev)
    fix = torch.zeros((N, C, h, w), dtype=torch.int64, device=dev)
    dzq = torch.empty_like(zq)
    lib = _lib.load()
    with torch.cuda.device(dev):
        st = _stream(zq)
        if weights is None:
            _lib.check(lib.mas_loss_scales(acc.data_ptr(), grad_out.data_ptr(), flags, scale.data_ptr(), st), "mas_loss_scales")
        else:               # grad_out is dL/d(total) [1]; the chain rule through the weighted sum happens in the kernel
            _lib.check(lib.mas_loss_scales_weighted(acc.data_ptr(), grad_out.data_ptr(), weights.data_ptr(), flags, scale.data_ptr(), st),
                       "mas_loss_scales_weighted")
        _lib.check(lib.mas_partial_loss_bwd_lowres(zq.data_ptr(), h, w, spx.data_ptr(), _id_code(spx), mask.data_ptr(), bits.data_ptr(),
                                                   gmax.data_ptr() if gmax is not None else None, scale.data_ptr(), N, C, H, W, S,
                                                   invT, flags, fix.data_ptr(), st), "mas_partial_loss_bwd_lowres")
        _lib.check(lib.mas_fix_to_float(fix.data_ptr(), fix.numel(), _lib.GRAD_FRAC, dzq.data_ptr(), st), "mas_fix_to_float")
    return (dzq, fix) if want_fix else dzq


def path_ranks(paths):
    """Rank of every image's joined path string in ascending (Python ``str``) order -- the tie-break
    the reference's tuple sort applies after the score (``active_selection/base.py:37``).
    Returns (img_rank[n], img_of_rank[n]) as int32 numpy arrays."""
    order = sorted(range(len(paths)), key=lambda i: paths[i])
    img_of_rank = np.asarray(order, dtype=np.int32)
    img_rank = np.empty(len(paths), dtype=np.int32)
    img_rank[img_of_rank] = np.arange(len(paths), dtype=np.int32)
    return img_rank, img_of_rank


def region_keys(scores, valid, img_rank):
    """64-bit sort keys [n_img*S] (int64 bits of uint64) for scores [n_img,S] f32."""
    _need(scores, "scores", torch.float32)
    _need(img_rank, "img_rank", torch.int32)
    n_img, S = scores.shape
    if valid is not None:
        valid = _mask_u8(valid)
        if tuple(valid.shape) != (n_img, S):
            raise ValueError("valid must be [n_img, S]")
    keys = torch.empty(n_img * S, dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().mas_region_keys(scores.data_ptr(), valid.data_ptr() if valid is not None else None,
                                               img_rank.data_ptr(), n_img, S, keys.data_ptr(), _stream(scores)),
                   "mas_region_keys")
    return keys


def sort_keys_desc(keys):
    _need(keys, "keys", torch.int64)
    n = keys.numel()
    lib = _lib.load()
    ws = torch.empty(lib.mas_select_workspace_bytes(n), dtype=torch.uint8, device=keys.device)
    out = torch.empty_like(keys)
    with torch.cuda.device(keys.device):
        _lib.check(lib.mas_sort_keys_desc(keys.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(keys)),
                   "mas_sort_keys_desc")
    return out


def budget_walk(sorted_keys, region_cost, img_of_rank, S, budget, max_out=None):
    """Returns (n_selected int64[1] on device, sel_img i32[max_out], sel_id i32[max_out], sel_score f32[max_out]).
    ``region_cost`` u8 [n_img*S] (``multi_hot_cls[img, id].sum()``) or None for unit cost."""
    _need(sorted_keys, "sorted_keys", torch.int64)
    _need(img_of_rank, "img_of_rank", torch.int32)
    cost_bits = region_cost
    if cost_bits is not None:
        _need(cost_bits, "region_cost", torch.uint8)
    n = sorted_keys.numel()
    # every region costs >= 1 when costs are popcounts of non-empty rows; budget+1 outputs always suffice then
    max_out = n if max_out is None else min(n, max_out)
    dev = sorted_keys.device
    lib = _lib.load()
    ws = torch.empty(lib.mas_select_workspace_bytes(n), dtype=torch.uint8, device=dev)
    nsel = torch.empty(1, dtype=torch.int64, device=dev)
    sel_img = torch.empty(max_out, dtype=torch.int32, device=dev)
    sel_id = torch.empty(max_out, dtype=torch.int32, device=dev)
    sel_score = torch.empty(max_out, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mas_budget_walk(sorted_keys.data_ptr(), n, cost_bits.data_ptr() if cost_bits is not None else None,
                                       img_of_rank.data_ptr(), S, int(budget), max_out, nsel.data_ptr(), sel_img.data_ptr(),
                                       sel_id.data_ptr(), sel_score.data_ptr(), ws.data_ptr(), ws.numel(), _stream(sorted_keys)),
                   "mas_budget_walk")
    return nsel, sel_img, sel_id, sel_score


def minmax_normalize_(scores):
    """In place: (u - min(u[u != 0])) / max(...) over all region scores -- my_bvsb.py:79-81."""
    _need(scores, "scores", torch.float32)
    scratch = torch.empty(2, dtype=torch.int32, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().mas_minmax_normalize(scores.data_ptr(), scores.numel(), scratch.data_ptr(), _stream(scores)),
                   "mas_minmax_normalize")
    return scores


# ------------------------------------------------------------------------------------------------
# mIoU counters
# ------------------------------------------------------------------------------------------------
def iou_counts(outputs, outputs_all, targets, num_classes, ignore_label, counts=None):
    """Accumulate seen/correct/positive (+ the 3 "undefined"-class counters) into ``counts`` int64[3C+3].
    Reference: utils/miou.py:23-38, utils/miou_evalignore.py:20-32."""
    _need(targets, "targets", torch.int64)
    if outputs is not None:
        _need(outputs, "outputs", torch.int64)
    if outputs_all is not None:
        _need(outputs_all, "outputs_all", torch.int64)
    if counts is None:
        counts = torch.zeros(3 * num_classes + 3, dtype=torch.int64, device=targets.device)
    with torch.cuda.device(targets.device):
        _lib.check(_lib.load().mas_iou_counts(outputs.data_ptr() if outputs is not None else None,
                                              outputs_all.data_ptr() if outputs_all is not None else None,
                                              targets.data_ptr(), targets.numel(), num_classes, int(ignore_label),
                                              counts.data_ptr(), _stream(targets)), "mas_iou_counts")
    return counts


def logits_iou_counts(z, targets, num_classes, ignore_label, counts=None):
    """Fused argmax + counters from logits [B,channels,H,W] (channels = num_classes or num_classes+1)."""
    _need(z, "logits", torch.float32)
    _need(targets, "targets", torch.int64)
    B, CH, H, W = z.shape
    if tuple(targets.shape) != (B, H, W):
        raise ValueError("targets %s do not match logits %s" % (tuple(targets.shape), tuple(z.shape)))
    if counts is None:
        counts = torch.zeros(3 * num_classes + 3, dtype=torch.int64, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().mas_logits_iou_counts(z.data_ptr(), targets.data_ptr(), B, CH, H, W, num_classes,
                                                     int(ignore_label), counts.data_ptr(), _stream(z)), "mas_logits_iou_counts")
    return counts


# ------------------------------------------------------------------------------------------------
# single-pass acquisition scan
# ------------------------------------------------------------------------------------------------
def weights_to_fixed31(cls_w):
    """floor(w * 2^31) as int32-viewed uint32 (host, exact): the integer class weights of the single-pass
    finalize.  ``cls_w``: float32 numpy array or None (-> all ones)."""
    w = np.asarray(cls_w, dtype=np.float32).astype(np.float64)
    return np.floor(w * 2147483648.0).astype(np.uint32)


def class_weight(prob_sum, hw, batch_size, n_batches, coeff):
    """Device-side class weights from the per-picture class sums [n_img, C] int64 (``mas_class_weight``): returns
    (cum f64 [C], cls_w f32 [C], w31 int32 [C] = floor(cls_w * 2^31) bits).  No host synchronisation."""
    _need(prob_sum, "prob_sum", torch.int64)
    n_img, C = prob_sum.shape
    dev = prob_sum.device
    cum = torch.empty(C, dtype=torch.float64, device=dev)
    w = torch.empty(C, dtype=torch.float32, device=dev)
    w31 = torch.empty(C, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mas_class_weight(prob_sum.data_ptr(), n_img, C, int(hw), int(batch_size), int(n_batches),
                                                float(coeff), cum.data_ptr(), w.data_ptr(), w31.data_ptr(), _stream(prob_sum)),
                   "mas_class_weight")
    return cum, w, w31


def single_pass_accum(z, spx, S, invT, prob_sum=None, class_sum=None, hist=None):
    """One scan: (prob_sum [B,C] i64, class_sum [B,S,C] i64, hist [B,S,C] i32), all accumulated into."""
    _need(z, "z", torch.float32)
    _need(spx, "spx")
    B, C, H, W = z.shape
    if tuple(spx.shape) != (B, H, W):
        raise ValueError("spx shape %s does not match logits %s" % (tuple(spx.shape), tuple(z.shape)))
    dev = z.device
    if prob_sum is None:
        prob_sum = torch.zeros((B, C), dtype=torch.int64, device=dev)
    if class_sum is None:
        class_sum = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
    if hist is None:
        hist = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mas_single_pass_accum(z.data_ptr(), spx.data_ptr(), _id_code(spx), B, C, H, W, S, invT,
                                                     prob_sum.data_ptr(), class_sum.data_ptr(), hist.data_ptr(), _stream(z)),
                   "mas_single_pass_accum")
    return prob_sum, class_sum, hist


def single_pass_accum_lowres(zq, size, spx, S, invT, prob_sum=None, class_sum=None, hist=None, generic=False):
    """``single_pass_accum`` of ``F.interpolate(zq, size, 'bilinear', align_corners=False)`` without materialising it: ``zq``
    [B,C,h,w] quarter-resolution logits, ``spx`` [B,H,W]; same outputs, bit for bit.  ``generic`` (tests / A-B measurements): keep
    the generic tap reads at the exact x4 ratio too (a per-call flag; the results are bit-identical either way)."""
    _need(zq, "zq", torch.float32)
    _need(spx, "spx")
    B, C, h, w = zq.shape
    H, W = int(size[0]), int(size[1])
    if tuple(spx.shape) != (B, H, W):
        raise ValueError("spx shape %s does not match logits %s at size %s" % (tuple(spx.shape), tuple(zq.shape), (H, W)))
    dev = zq.device
    if prob_sum is None:
        prob_sum = torch.zeros((B, C), dtype=torch.int64, device=dev)
    if class_sum is None:
        class_sum = torch.zeros((B, S, C), dtype=torch.int64, device=dev)
    if hist is None:
        hist = torch.zeros((B, S, C), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mas_single_pass_accum_lowres_opt(zq.data_ptr(), h, w, spx.data_ptr(), _id_code(spx), B, C, H, W, S, invT,
                                                                prob_sum.data_ptr(), class_sum.data_ptr(), hist.data_ptr(),
                                                                _lib.LOWRES_GENERIC if generic else 0, _stream(zq)),
                   "mas_single_pass_accum_lowres_opt")
    return prob_sum, class_sum, hist


def region_finalize_weighted(class_sum, hist, w31, ban_class=-1, want_hist_i64=False):
    """Weighted mean per region from the single-pass accumulators.  ``w31``: int32 tensor [C] holding uint32 bits."""
    _need(class_sum, "class_sum", torch.int64)
    _need(hist, "hist", torch.int32)
    _need(w31, "w31", torch.int32)
    C = hist.shape[-1]
    shape = hist.shape[:-1]
    n = hist.numel() // C
    dev = hist.device
    score = torch.empty(shape, dtype=torch.float32, device=dev)
    dom = torch.empty(shape, dtype=torch.int32, device=dev)
    cnt = torch.empty(shape, dtype=torch.int32, device=dev)
    h64 = torch.empty(hist.shape, dtype=torch.int64, device=dev) if want_hist_i64 else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mas_region_finalize_weighted(
            class_sum.data_ptr(), hist.data_ptr(), n, C, w31.data_ptr(), ban_class, score.data_ptr(), dom.data_ptr(),
            cnt.data_ptr(), h64.data_ptr() if h64 is not None else None, _stream(hist)), "mas_region_finalize_weighted")
    return score, dom, cnt, h64


def region_reweight_(scores, dominant, ban_class=-1, cls_w=None):
    """In place: zero the regions whose dominant class is ``ban_class`` and multiply by ``cls_w[dominant]``
    (my_bvsb_banignore.py:58-61, my_bvsb_clsbal_v2_banignore.py:60-74)."""
    _need(scores, "scores", torch.float32)
    _need(dominant, "dominant", torch.int32)
    if cls_w is not None:
        _need(cls_w, "cls_w", torch.float32)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().mas_region_reweight(scores.data_ptr(), dominant.data_ptr(), scores.numel(), ban_class,
                                                   cls_w.data_ptr() if cls_w is not None else None, _stream(scores)),
                   "mas_region_reweight")
    return scores


def dominant_hist(dominant, C):
    """int64[C]: number of regions per dominant class."""
    _need(dominant, "dominant", torch.int32)
    counts = torch.zeros(C, dtype=torch.int64, device=dominant.device)
    with torch.cuda.device(dominant.device):
        _lib.check(_lib.load().mas_dominant_hist(dominant.data_ptr(), dominant.numel(), C, counts.data_ptr(), _stream(dominant)),
                   "mas_dominant_hist")
    return counts


# ------------------------------------------------------------------------------------------------
# K9: stage-2 cosine pseudo labels
# ------------------------------------------------------------------------------------------------
def stage2_pseudo_labels(feats, logits, targets, spmasks, superpixels, include_onehot=True):
    """Pseudo-label maps int64 [N,H,W] (255 = none) -- trainer/eval_save_cosplbl_prop[_includeonehot].py:121-314.

    feats [N,Ch,fh,fw]: the L2-normalised point features of ``feat_forward`` BEFORE the bilinear upsampling (full
    resolution is accepted too); logits [N,C,H,W]; targets u8 [N,S,C]; spmasks bool [N,H,W]; superpixels int64.
    The small index bookkeeping (prototype list, per-prototype medians) uses torch ops on the device; feature
    interpolation, similarities, adjacency and propagation are HIP kernels."""
    _need(feats, "feats", torch.float32)
    _need(logits, "logits", torch.float32)
    _need(superpixels, "superpixels", torch.int64)
    mask = _mask_u8(spmasks)
    if targets.dtype != torch.uint8:
        targets = targets.to(torch.uint8)
    N, C, H, W = logits.shape
    Ch, fh, fw = feats.shape[1:]
    S = targets.shape[1]
    dev = logits.device
    lib = _lib.load()
    bits = target_bits(targets.contiguous())
    flags = _lib.LOSS_GROUP | (0 if include_onehot else _lib.LOSS_GROUP_ONLY_MULTI)
    out = torch.full((N, H, W), 255, dtype=torch.int64, device=dev)
    words = (S + 31) // 32
    with torch.cuda.device(dev):
        st = _stream(logits)
        for i in range(N):
            _, _, gmax = partial_loss_fwd(logits[i:i + 1], superpixels[i:i + 1], mask[i:i + 1], bits[i:i + 1], 1.0, flags)
            g = gmax[0]
            nz = (g != 0).nonzero()                              # ordered by (superpixel, class)
            n_proto = nz.shape[0]
            if n_proto == 0:
                continue
            proto_s, proto_c = nz[:, 0], nz[:, 1]
            proto_pix = (0xffffffff - (g[proto_s, proto_c] & 0xffffffff)).to(torch.int32)
            counts = torch.bincount(proto_s, minlength=S)
            p_start = torch.zeros(S + 1, dtype=torch.int32, device=dev)
            p_start[1:] = torch.cumsum(counts, 0).to(torch.int32)
            p_cls = proto_c.to(torch.int32).contiguous()
            P = torch.empty((n_proto, Ch), dtype=torch.float32, device=dev)
            f, sp, mk = feats[i], superpixels[i], mask[i]
            _lib.check(lib.mas_stage2_gather_protos(f.data_ptr(), Ch, fh, fw, H, W, proto_pix.data_ptr(), n_proto, P.data_ptr(), st),
                       "mas_stage2_gather_protos")
            nn = torch.empty(H * W, dtype=torch.int32, device=dev)
            nn_sim = torch.empty(H * W, dtype=torch.float32, device=dev)
            _lib.check(lib.mas_stage2_assign(f.data_ptr(), Ch, fh, fw, H, W, sp.data_ptr(), mk.data_ptr(), S, p_start.data_ptr(),
                                             P.data_ptr(), nn.data_ptr(), nn_sim.data_ptr(), st), "mas_stage2_assign")
            # per-prototype lower median of the assigned similarities (1.0 when a prototype attracts no pixel)
            sel = (nn >= 0).nonzero().squeeze(1)
            thr = torch.ones(n_proto, dtype=torch.float32, device=dev)
            if sel.numel():
                pid, sim = nn[sel].long(), nn_sim[sel]
                order = torch.argsort(sim, stable=True)
                order = order[torch.argsort(pid[order], stable=True)]            # sorted by (prototype, similarity)
                pid_s, sim_s = pid[order], sim[order]
                cnt = torch.bincount(pid_s, minlength=n_proto)
                first = torch.cumsum(cnt, 0) - cnt
                has = cnt > 0
                thr[has] = sim_s[(first + (cnt - 1) // 2)[has]]
            adj = torch.zeros((S, words), dtype=torch.int32, device=dev)
            _lib.check(lib.mas_stage2_adjacency(sp.data_ptr(), H, W, S, p_start.data_ptr(), adj.data_ptr(), st), "mas_stage2_adjacency")
            _lib.check(lib.mas_stage2_propagate(f.data_ptr(), Ch, fh, fw, H, W, sp.data_ptr(), S, adj.data_ptr(), p_start.data_ptr(),
                                                p_cls.data_ptr(), P.data_ptr(), thr.data_ptr(), nn.data_ptr(), out[i].data_ptr(), st),
                       "mas_stage2_propagate")
    return out


# ------------------------------------------------------------------------------------------------
# K7: ASPP depthwise triple
# ------------------------------------------------------------------------------------------------
class _AsppDepthwise3(torch.autograd.Function):
    """(y6, y12, y18) = three dilated depthwise 3x3 convolutions of the same map, one read of x
    (models/segmentation/deeplabv3.py:174-177 x 3 branches)."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2, d0, d1, d2):
        x = x.contiguous()
        w0, w1, w2 = w0.contiguous(), w1.contiguous(), w2.contiguous()
        N, C, H, W = x.shape
        ys = [torch.empty_like(x) for _ in range(3)]
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_aspp_dw3_fwd(x.data_ptr(), w0.data_ptr(), w1.data_ptr(), w2.data_ptr(), N, C, H, W, d0, d1, d2,
                                                    ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), _stream(x)), "mas_aspp_dw3_fwd")
        ctx.save_for_backward(x, w0, w1, w2)
        ctx.dil = (d0, d1, d2)
        return tuple(ys)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        x, w0, w1, w2 = ctx.saved_tensors
        d0, d1, d2 = ctx.dil
        N, C, H, W = x.shape
        g0, g1, g2 = g0.contiguous(), g1.contiguous(), g2.contiguous()
        lib = _lib.load()
        dx = dws = None
        with torch.cuda.device(x.device):
            st = _stream(x)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(lib.mas_aspp_dw3_bwd_x(g0.data_ptr(), g1.data_ptr(), g2.data_ptr(), w0.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                                  N, C, H, W, d0, d1, d2, dx.data_ptr(), st), "mas_aspp_dw3_bwd_x")
            if any(ctx.needs_input_grad[1:4]):
                dws = [torch.empty_like(w0), torch.empty_like(w1), torch.empty_like(w2)]
                _lib.check(lib.mas_aspp_dw3_bwd_w(x.data_ptr(), g0.data_ptr(), g1.data_ptr(), g2.data_ptr(), N, C, H, W, d0, d1, d2,
                                                  dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), st), "mas_aspp_dw3_bwd_w")
        dws = dws or [None, None, None]
        return dx, dws[0], dws[1], dws[2], None, None, None


def aspp_depthwise3(x, w0, w1, w2, dilations):
    _need(x, "x", torch.float32)
    for w in (w0, w1, w2):
        if tuple(w.shape) != (x.shape[1], 1, 3, 3):
            raise ValueError("depthwise weights must be [C,1,3,3]")
    return _AsppDepthwise3.apply(x, w0, w1, w2, int(dilations[0]), int(dilations[1]), int(dilations[2]))


class _Depthwise3x3(torch.autograd.Function):
    """y = depthwise 3x3 (stride 1, padding = dilation, no bias) of x (models/segmentation/deeplabv3.py:174-177)."""

    @staticmethod
    def forward(ctx, x, w, d):
        x, w = x.contiguous(), w.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_depthwise3x3_fwd(x.data_ptr(), w.data_ptr(), N, C, H, W, d, y.data_ptr(), _stream(x)),
                       "mas_depthwise3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.dil = d
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        d = ctx.dil
        N, C, H, W = x.shape
        g = g.contiguous()
        lib = _lib.load()
        dx = dw = None
        with torch.cuda.device(x.device):
            st = _stream(x)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(lib.mas_depthwise3x3_bwd_x(g.data_ptr(), w.data_ptr(), N, C, H, W, d, dx.data_ptr(), st), "mas_depthwise3x3_bwd_x")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w)
                part = torch.empty((N, C, 9), dtype=torch.float32, device=x.device)
                _lib.check(lib.mas_depthwise3x3_bwd_w(x.data_ptr(), g.data_ptr(), N, C, H, W, d, part.data_ptr(), dw.data_ptr(), st),
                           "mas_depthwise3x3_bwd_w")
        return dx, dw, None


def depthwise3x3_supported(x, dilation):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (16 + 2 * dilation) * (x.shape[3] + 2 * dilation) * 4 <= 64 * 1024


def depthwise3x3(x, w, dilation):
    _need(x, "x", torch.float32)
    if tuple(w.shape) != (x.shape[1], 1, 3, 3):
        raise ValueError("depthwise weight must be [C,1,3,3]")
    return _Depthwise3x3.apply(x, w, int(dilation))


class _UpsampleBilinear(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) with a deterministic gather backward."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = x.contiguous()
        N, C, Hi, Wi = x.shape
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_upsample_bilinear_fwd(x.data_ptr(), N * C, Hi, Wi, Ho, Wo, y.data_ptr(), _stream(x)),
                       "mas_upsample_bilinear_fwd")
        ctx.shape = (N, C, Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, g):
        N, C, Hi, Wi, Ho, Wo = ctx.shape
        g = g.contiguous()
        gx = torch.empty((N, C, Hi, Wi), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().mas_upsample_bilinear_bwd(g.data_ptr(), N * C, Hi, Wi, Ho, Wo, gx.data_ptr(), _stream(g)),
                       "mas_upsample_bilinear_bwd")
        return gx, None, None


def upsample_bilinear_supported(x, size):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and int(size[1]) <= 6 * x.shape[3] and int(size[0]) >= x.shape[2]
            and int(size[1]) >= x.shape[3] and x.shape[0] * x.shape[1] <= 65535 and int(size[0]) <= 65535)


def upsample_bilinear(x, size):
    _need(x, "x", torch.float32)
    return _UpsampleBilinear.apply(x, int(size[0]), int(size[1]))


def upsample_bilinear_into(x, out):
    """Inference only: write the upsampling of x [N,C,hi,wi] into ``out`` [N,C,Ho,Wo], a view whose per-picture [C,Ho,Wo]
    blocks are contiguous (e.g. a channel slice of a concatenation buffer) -- one launch per picture, no copy afterwards."""
    _need(x, "x", torch.float32)
    N, C, Hi, Wi = x.shape
    Ho, Wo = out.shape[2], out.shape[3]
    if out.shape[0] != N or out.shape[1] != C or out.stride(3) != 1 or out.stride(2) != Wo or out.stride(1) != Ho * Wo:
        raise ValueError("out must be [N,C,Ho,Wo] with contiguous per-picture blocks")
    lib = _lib.load()
    with torch.cuda.device(x.device):
        for n in range(N):
            _lib.check(lib.mas_upsample_bilinear_fwd(x[n].data_ptr(), C, Hi, Wi, Ho, Wo, out[n].data_ptr(), _stream(x)),
                       "mas_upsample_bilinear_fwd")
    return out


# ------------------------------------------------------------------------------------------------
# BatchNorm2d + ReLU + residual add, fused (csrc/bn.hip)
# ------------------------------------------------------------------------------------------------
def _opt(t):
    return t.data_ptr() if t is not None else None


_BN_COUNTERS = {}


def _bn_counters(dev, C):
    """The per-(device, stream) completion counters of the last-block form of the BatchNorm passes (one zeroed word per channel; the
    kernels leave them zeroed) under MAS_BN_LASTBLOCK=on.  Default off: None -- the statistics come from their own launch.  (Round 5
    measured the last-block form at 25.7-25.9 ms per training step against 25.6 for the three-launch form: the store-drain-publish
    tail of every workgroup costs what the 115 five-microsecond launches did.  Kept for the A/B and its bit-identity test.)"""
    if os.environ.get("MAS_BN_LASTBLOCK", "off") != "on":
        return None
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _BN_COUNTERS.get(key)
    if buf is None or buf.numel() < C:
        buf = _BN_COUNTERS[key] = torch.zeros(max(4096, C), dtype=torch.int32, device=dev)
    return buf


class _BNActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, num_batches_tracked, eps, momentum, relu, partials=None):
        x = x.contiguous()
        N, C, H, W = x.shape
        HW = H * W
        dev = x.device
        res = residual.contiguous() if residual is not None else None
        lib = _lib.load()
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        # ReLU mask, one byte per aligned group of four outputs: the backward then reads 1/16 of the bytes of y
        mask = torch.empty(int(lib.mas_bn_mask_bytes(N, C, HW)), dtype=torch.uint8, device=dev) if relu else None
        with torch.cuda.device(dev):
            if partials is not None:
                # (sum, sum of squares) per channel and pixel set, formed in the epilogue of the convolution that produced x
                # (conv_sk(..., stats=True)): no reduction pass over x
                if partials.dtype != torch.float64 or partials.dim() != 3 or partials.shape[0] != C or partials.shape[2] != 2:
                    raise ValueError("partials must be [C, slots, 2] float64")
                _lib.check(lib.mas_bn_act_train_fwd_stats(x.data_ptr(), partials.data_ptr(), int(partials.shape[1]), _opt(weight), _opt(bias), _opt(res),
                                                          N, C, HW, float(eps), float(momentum), int(relu), _opt(running_mean), _opt(running_var),
                                                          _opt(num_batches_tracked), mean.data_ptr(), invstd.data_ptr(), y.data_ptr(), _opt(mask),
                                                          _stream(x)), "mas_bn_act_train_fwd_stats")
            else:
                ws = torch.empty(int(lib.mas_bn_workspace_bytes(N, C, HW)), dtype=torch.uint8, device=dev)
                _lib.check(lib.mas_bn_act_train_fwd(x.data_ptr(), _opt(weight), _opt(bias), _opt(res), N, C, HW, float(eps), float(momentum),
                                                    int(relu), _opt(running_mean), _opt(running_var), _opt(num_batches_tracked),
                                                    mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), y.data_ptr(), _opt(mask), _opt(_bn_counters(dev, C)),
                                                    _stream(x)),
                           "mas_bn_act_train_fwd")
        # the kernel updated the running statistics through raw pointers: bump their version counters as an in-place
        # torch op would, so caches keyed on (data_ptr, _version) -- _conv1x1_constants -- see the change
        for buf in (running_mean, running_var, num_batches_tracked):
            if buf is not None:
                torch.autograd.graph.increment_version(buf)
        ctx.save_for_backward(x, y if mask is None else None, weight, mean, invstd, mask)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd, mask = ctx.saved_tensors
        N, C, H, W = x.shape
        HW = H * W
        dev = x.device
        dy = dy.contiguous()
        lib = _lib.load()
        ws = torch.empty(int(lib.mas_bn_workspace_bytes(N, C, HW)), dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        dg = torch.empty(C, dtype=torch.float32, device=dev) if (weight is not None and ctx.needs_input_grad[1]) else None
        db = torch.empty(C, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            _lib.check(lib.mas_bn_act_train_bwd(dy.data_ptr(), x.data_ptr(), _opt(y), _opt(mask), _opt(weight), mean.data_ptr(), invstd.data_ptr(),
                                                N, C, HW, int(ctx.relu), ws.data_ptr(), dx.data_ptr(), _opt(dres), _opt(dg), _opt(db),
                                                _opt(_bn_counters(dev, C)), _stream(x)), "mas_bn_act_train_bwd")
        if ctx.has_res and dres is None and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dg, db, dres, None, None, None, None, None, None, None


def bn_act_supported(bn, x, residual=None):
    """Fused path: f32 NCHW GPU tensors; training mode (batch statistics) or inference without autograd.  A frozen
    BatchNorm (eval mode with gradients flowing through it) stays on the PyTorch ops."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] <= 65535 and x.shape[1] <= 65535):
        return False
    if residual is not None and (residual.shape != x.shape or residual.dtype != torch.float32):
        return False
    if bn.training:
        return bn.momentum is not None and x.shape[0] * x.shape[2] * x.shape[3] > 1
    if not bn.track_running_stats:
        return False
    return not (torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)))


def bn_act(bn, x, relu=True, residual=None, partials=None):
    """relu?(bn(x) + residual) for a ``torch.nn.BatchNorm2d`` module ``bn`` (parameters, running statistics and
    ``num_batches_tracked`` are the module's own and are updated as PyTorch updates them).  ``partials`` (training): the
    per-channel partial sums of x that conv_train(..., stats=True) returns beside x."""
    if bn.training:
        rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else (None, None, None)
        return _BNActTrain.apply(x, bn.weight, bn.bias, residual, rm, rv, nbt, bn.eps, bn.momentum, relu, partials)
    x = x.contiguous()
    N, C, H, W = x.shape
    res = residual.contiguous() if residual is not None else None
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_bn_act_eval_fwd(x.data_ptr(), _opt(bn.weight), _opt(bn.bias), bn.running_mean.data_ptr(),
                                                   bn.running_var.data_ptr(), _opt(res), N, C, H * W, float(bn.eps), int(relu), y.data_ptr(),
                                                   _stream(x)), "mas_bn_act_eval_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# the 1x1 convolution of a 1x1 map (ASPP image-pooling branch): a fixed-order [N,K] x [K,M] product (csrc/head.hip)
# ------------------------------------------------------------------------------------------------
class _DenseSmall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = x.contiguous(), w.contiguous()
        N, K = x.shape
        M = w.shape[0]
        y = torch.empty((N, M), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_dense_small_fwd(x.data_ptr(), w.data_ptr(), N, K, M, y.data_ptr(), _stream(x)), "mas_dense_small_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, K = x.shape
        M = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_dense_small_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), N, K, M, _opt(dx), _opt(dw), _stream(x)),
                       "mas_dense_small_bwd")
        return dx, dw


def conv1x1_on_1x1_supported(conv, x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] * x.shape[3] == 1 and conv.kernel_size == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and x.shape[1] == conv.in_channels
            and x.shape[0] * conv.out_channels <= 1 << 20)


def conv1x1_on_1x1(conv, x):
    """conv(x) for a 1x1 convolution of a 1x1 map [N,K,1,1] -> [N,M,1,1] with autograd, fixed summation order (mas_dense_small_*)."""
    y = _DenseSmall.apply(x.flatten(1), conv.weight.flatten(1))
    return y[:, :, None, None]


# ------------------------------------------------------------------------------------------------
# K8: cosine classifier (csrc/head.hip)
# ------------------------------------------------------------------------------------------------
class _CosineHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, proxy_hat):
        feat, proxy_hat = feat.contiguous(), proxy_hat.contiguous()
        N, Ch, H, W = feat.shape
        K = proxy_hat.shape[0]
        logits = torch.empty((N, K, H, W), dtype=torch.float32, device=feat.device)
        inv = torch.empty((N, H, W), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(_lib.load().mas_cosine_head_fwd(feat.data_ptr(), proxy_hat.data_ptr(), N, Ch, K, H * W, 1e-12, logits.data_ptr(),
                                                       inv.data_ptr(), _stream(feat)), "mas_cosine_head_fwd")
        ctx.save_for_backward(feat, proxy_hat, logits, inv)
        return logits

    @staticmethod
    def backward(ctx, g):
        feat, proxy_hat, logits, inv = ctx.saved_tensors
        N, Ch, H, W = feat.shape
        K = proxy_hat.shape[0]
        g = g.contiguous()
        dfeat = dph = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.empty_like(feat)
            with torch.cuda.device(feat.device):
                _lib.check(_lib.load().mas_cosine_head_bwd(feat.data_ptr(), proxy_hat.data_ptr(), logits.data_ptr(), inv.data_ptr(),
                                                           g.data_ptr(), N, Ch, K, H * W, dfeat.data_ptr(), _stream(feat)),
                           "mas_cosine_head_bwd")
        if ctx.needs_input_grad[1]:
            gs = (g * inv[:, None]).reshape(N, K, H * W)                       # small: [N,K,HW]
            dph = torch.bmm(gs, feat.reshape(N, Ch, H * W).transpose(1, 2)).sum(0)      # [K,Ch] = sum_n gs_n @ f_n^T  (hipBLASLt)
        return dfeat, dph


def cosine_head_supported(feat, proxy):
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 4 and proxy.dim() == 4 and proxy.shape[2:] == (1, 1)
            and proxy.shape[0] in (19, 20, 21) and proxy.shape[1] == feat.shape[1] and feat.shape[0] <= 65535)


def cosine_head(feat, proxy):
    """logits = conv2d(normalize(feat), normalize(proxy)) (deeplabv3.py:121-124); proxy [K,Ch,1,1] raw weights (their
    normalisation over dim 1 stays a PyTorch op on the [K,Ch] tensor, so its gradient is autograd's)."""
    phat = torch.nn.functional.normalize(proxy, dim=1).reshape(proxy.shape[0], proxy.shape[1])
    return _CosineHead.apply(feat, phat)


# ------------------------------------------------------------------------------------------------
# stem max-pool (csrc/pool.hip)
# ------------------------------------------------------------------------------------------------
class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        arg = torch.empty((N, C, Ho, Wo), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mas_maxpool3s2_fwd(x.data_ptr(), N * C, H, W, y.data_ptr(), arg.data_ptr(), _stream(x)), "mas_maxpool3s2_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        g = g.contiguous()
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().mas_maxpool3s2_bwd(g.data_ptr(), arg.data_ptr(), N * C, H, W, dx.data_ptr(), _stream(g)), "mas_maxpool3s2_bwd")
        return dx


def maxpool3s2_supported(pool, x):
    def two(v):
        return (v, v) if isinstance(v, int) else tuple(v)
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] * x.shape[1] <= 65535 and x.shape[2] <= 65535
            and two(pool.kernel_size) == (3, 3) and two(pool.stride) == (2, 2) and two(pool.padding) == (1, 1)
            and two(pool.dilation) == (1, 1) and not pool.ceil_mode and not pool.return_indices)


def maxpool3s2(x):
    _need(x, "x", torch.float32)
    return _MaxPool3s2.apply(x)


# ------------------------------------------------------------------------------------------------
# small-K 1x1 convolution with the inference BatchNorm + ReLU epilogue (csrc/conv1x1.hip)
# ------------------------------------------------------------------------------------------------
def conv1x1_bn_act_supported(conv, bn, x):
    """Inference only (eval-mode BatchNorm, no autograd): 1x1 / stride 1 / no bias, at most 64 output channels -- the
    shapes where the HIP kernel beats MIOpen's GEMM path (1.6-2.2x at 256x512; tools/conv1x1_hip_probe.py)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not bn.training and bn.track_running_stats
            and not torch.is_grad_enabled() and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and conv.out_channels in (32, 64) and conv.in_channels % 4 == 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] <= 65535)


def _conv1x1_constants(conv, bn):
    """(transposed weight [K,M], scale [M], shift [M]) cached on the conv module until a parameter / statistic changes."""
    tensors = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = _versions(tensors)
    cache = getattr(conv, '_mas_conv1x1_cache', None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            M, K = conv.out_channels, conv.in_channels
            w_t = conv.weight.reshape(M, K).t().contiguous()
            inv = torch.rsqrt(bn.running_var.double() + bn.eps)
            g = bn.weight.double() if bn.weight is not None else torch.ones_like(inv)
            b = bn.bias.double() if bn.bias is not None else torch.zeros_like(inv)
            scale = (g * inv)
            shift = (b - bn.running_mean.double() * scale)
            cache = (key, w_t, scale.float().contiguous(), shift.float().contiguous())
        conv._mas_conv1x1_cache = cache
    return cache[1:]


def conv1x1_bn_act(conv, bn, x, relu=True, residual=None):
    """relu?(bn(conv(x)) + residual) in one kernel (inference)."""
    x = x.contiguous()
    N, K, H, W = x.shape
    M = conv.out_channels
    w_t, scale, shift = _conv1x1_constants(conv, bn)
    res = residual.contiguous() if residual is not None else None
    y = torch.empty((N, M, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_conv1x1_fwd(x.data_ptr(), w_t.data_ptr(), N, K, M, H * W, scale.data_ptr(), shift.data_ptr(), _opt(res),
                                               int(relu), y.data_ptr(), _stream(x)), "mas_conv1x1_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# dense convolutions on the f32 matrix cores (csrc/conv_mfma.hip)
# ------------------------------------------------------------------------------------------------
def conv_mfma_supported(conv, x):
    """Shapes the implicit-GEMM MFMA kernel takes: 1x1 / 3x3, stride 1 or 2 (3x3 stride 2 only undilated), padding =
    dilation for 3x3, no groups, no bias, Cin % 8 (3x3) / % 16 (1x1) == 0 (any Cout: padded to 64 inside), fp32 NCHW on the GPU."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and conv.bias is None):
        return False
    k, s, d, pd = conv.kernel_size, conv.stride, conv.dilation, conv.padding
    if k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or pd[0] != pd[1] or k[0] not in (1, 3) or s[0] not in (1, 2):
        return False
    if k[0] == 3 and (pd[0] != d[0] or d[0] > 2 or (s[0] == 2 and d[0] != 1)):
        return False
    if k[0] == 1 and (pd[0] != 0 or d[0] != 1):
        return False
    if x.shape[1] != conv.in_channels:
        return False
    return _lib.load().mas_conv_chunk(k[0], conv.in_channels) > 0 and conv.in_channels * x.shape[2] * x.shape[3] < 2 ** 31


# Parameter epoch: advanced after EVERY optimizer step of the process.  The version counter of a tensor is not enough to key a
# cache of derived constants on: fused optimizers (torch.optim.AdamW(fused=True), what trainer/base.py uses on the GPU) update
# parameters in place without bumping ``_version``, so a packed weight or a folded BatchNorm would silently go stale after the
# first training step.
_PARAM_EPOCH = [0]


def _bump_param_epoch(*_args, **_kwargs):
    _PARAM_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook
    _reg_step_hook(_bump_param_epoch)
except ImportError:                                     # pragma: no cover  (torch < 2.0)
    # Without the hook nothing tells this module that a fused optimizer stepped: packed weights and folded BatchNorm constants
    # would go stale after the first step and training would go on with wrong weights.  conv_train_plan then declines every layer
    # (the nn.Module / MIOpen path needs no derived constants) and says why, once.
    _reg_step_hook = None
    import warnings
    warnings.warn("torch.optim has no register_optimizer_step_post_hook: the package's training convolutions are disabled "
                  "(MIOpen is used); call ops.invalidate_parameter_caches() after every optimizer step to re-enable them by hand")


def invalidate_parameter_caches():
    """Call after changing parameters through raw pointers or anything else that bypasses both the version counters and
    torch.optim (every cache of packed weights / folded BatchNorm constants is rebuilt at its next use)."""
    _bump_param_epoch()


def _versions(tensors):
    return (_PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def _conv_packed_weight(conv):
    """The weight as mas_conv_fwd reads it -- [Cin/CK][KC/8][2][Cout][4]: k-step kk = tap * (CK/2) + cp pairs the input
    channels c = 2 cp + h (h = lane half of the MFMA), four consecutive k-steps of one (half, output channel) are
    adjacent -- cached on the module until the parameter changes."""
    key = _versions((conv.weight,))
    cache = getattr(conv, '_mas_conv_pack', None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            M, K, kh, kw = conv.weight.shape
            ck = _lib.load().mas_conv_chunk(kh, K)
            taps = kh * kw
            w = conv.weight.detach()
            if M % 64:                                              # output channels padded to 64 with zero rows (never stored)
                w = torch.cat([w, w.new_zeros((64 - M % 64, K, kh, kw))], dim=0)
                M = w.shape[0]
            w = w.reshape(M, K // ck, ck // 2, 2, taps)                             # [m, chunk, cp, h, tap]
            w = w.permute(1, 4, 2, 3, 0).reshape(K // ck, taps * ck // 2, 2, M)     # [chunk, kk = tap * ck/2 + cp, h, m]
            w = w.reshape(K // ck, taps * ck // 8, 4, 2, M).permute(0, 1, 3, 4, 2).contiguous()     # [chunk, q, h, m, j]
        cache = conv._mas_conv_pack = (key, w)
    return cache[1]


def _bn_fold(bn):
    """(scale, shift) f32 [C] of an inference BatchNorm, cached on the module until a parameter or running statistic
    changes (the training kernels bump the buffers' version counters, _BNActTrain.forward)."""
    key = _versions((bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cache = getattr(bn, '_mas_fold', None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var.double() + bn.eps)
            g = bn.weight.double() if bn.weight is not None else torch.ones_like(inv)
            b = bn.bias.double() if bn.bias is not None else torch.zeros_like(inv)
            scale = g * inv
            shift = b - bn.running_mean.double() * scale
        cache = bn._mas_fold = (key, scale.float().contiguous(), shift.float().contiguous())
    return cache[1], cache[2]


def conv_mfma(conv, x, bn=None, relu=False, residual=None):
    """relu?(bn(conv(x)) + residual) in one kernel; ``bn`` None -> the bare convolution (then residual / relu still apply)."""
    x = x.contiguous()
    N, K, H, W = x.shape
    M = conv.out_channels
    ks, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
    wt = _conv_packed_weight(conv)
    if bn is not None and bn.num_features != M:
        raise ValueError("BatchNorm has %d features, the convolution %d output channels" % (bn.num_features, M))
    scale, shift = _bn_fold(bn) if bn is not None else (None, None)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    if residual is not None and (tuple(residual.shape) != (N, M, Ho, Wo) or residual.dtype != torch.float32 or residual.device != x.device):
        raise ValueError("residual must be float32 %s on %s, got %s %s on %s"
                         % ((N, M, Ho, Wo), x.device, residual.dtype, tuple(residual.shape), residual.device))
    res = residual.contiguous() if residual is not None else None
    y = torch.empty((N, M, Ho, Wo), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_conv_fwd(x.data_ptr(), wt.data_ptr(), N, K, H, W, M, ks, s, d, _opt(scale), _opt(shift), _opt(res),
                                            int(relu), y.data_ptr(), _stream(x)), "mas_conv_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# the same convolutions on the bf16 matrix cores, f32 in / f32 out through exact three-term operand splits (csrc/conv_bx.hip)
# ------------------------------------------------------------------------------------------------
BX_ROLE_S2 = 2          # mas_conv_bx_pack role: the forward image of a 3x3 stride-2 convolution (nine shifted 1x1 stride-2 products)


def conv_bx_supported(conv, x):
    """Shapes mas_conv_bx_fwd takes: 1x1 at stride 1 (any plane) / 2 (H even, W % 8 == 0), 3x3 stride 1 with dilation 1 / 2
    (padding = dilation) on planes at least 32 wide, 3x3 stride 2 (H even, W % 8 == 0, input channels a multiple of 32); any other
    channel counts; no groups, no bias."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and conv.bias is None):
        return False
    if conv.stride[0] == 2 and x.data_ptr() % 16:
        return False
    if conv.stride[0] == 2 and conv.kernel_size[0] == 3 and os.environ.get("MAS_BX_S2K3", "on") == "off":      # (A/B: the f32 pipe)
        return False
    k, s, d, pd = conv.kernel_size, conv.stride, conv.dilation, conv.padding
    if k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or pd[0] != pd[1] or pd[0] != (d[0] if k[0] == 3 else 0) or x.shape[1] != conv.in_channels:
        return False
    return bool(_lib.load().mas_conv_bx_supported(conv.kernel_size[0], conv.stride[0], conv.dilation[0], conv.in_channels,
                                                  conv.out_channels, x.shape[2], x.shape[3]))


def _conv_bx_weight(conv):
    """The split weight image of mas_conv_bx_pack, cached on the module until the parameter changes (inference: once per
    checkpoint load)."""
    key = _versions((conv.weight,))
    cache = getattr(conv, '_mas_conv_bx_pack', None)
    if cache is None or cache[0] != key:
        lib = _lib.load()
        w = conv.weight.detach().contiguous()
        M, K, kh, _ = w.shape
        role = BX_ROLE_S2 if (kh == 3 and conv.stride[0] == 2) else 0        # (the strided 3x3: tap-major chunks of the 1x1 form)
        nbytes = lib.mas_conv_bx_packed_bytes(kh, K, M, role)
        if nbytes <= 0:
            raise ValueError("mas_conv_bx_pack does not take a %dx%d convolution with %d input channels" % (kh, kh, K))
        wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(lib.mas_conv_bx_pack(w.data_ptr(), None, M, K, kh, role, wp.data_ptr(), _stream(w)), "mas_conv_bx_pack")
        cache = conv._mas_conv_bx_pack = (key, wp)
    return cache[1]


def _conv_bx_folded(conv, bn):
    """(weight image with the inference BatchNorm's scale folded into its rows, shift [Cout]) for mas_conv_bx_fwd_dual, cached on the
    convolution until a parameter or running statistic changes."""
    key = _versions((conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cache = getattr(conv, '_mas_conv_bx_folded', None)
    if cache is None or cache[0] != key:
        scale, shift = _bn_fold(bn)
        cache = conv._mas_conv_bx_folded = (key, conv_bx_pack(conv.weight.detach(), 0, row_scale=scale), shift)
    return cache[1], cache[2]


def conv_bx_dual_supported(conv_a, xa, conv_b, xb):
    """Two 1x1 stride-1 convolutions with the same output channels on inputs of the same plane (conv3 and a stride-1 downsample of a
    Bottleneck), both shapes the split-bf16 kernel takes."""
    for conv, x in ((conv_a, xa), (conv_b, xb)):
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or not conv_bx_supported(conv, x):
            return False
    return conv_a.out_channels == conv_b.out_channels and xa.shape[0] == xb.shape[0] and xa.shape[2:] == xb.shape[2:]


def conv_bx_dual(conv_a, bn_a, xa, conv_b, bn_b, xb, relu=True):
    """relu?(bn_a(conv_a(xa)) + bn_b(conv_b(xb))) in ONE kernel at inference (mas_conv_bx_fwd_dual): both BatchNorm scales folded into
    the weight images, one accumulator set walks the channels of xa, then those of xb -- `out = relu(bn3(conv3(out)) + downsample(x))`
    of models/segmentation/backbone/resnet.py:143-160 without writing the identity branch to memory."""
    xa, xb = xa.contiguous(), xb.contiguous()
    N, Ka, H, W = xa.shape
    M = conv_a.out_channels
    wa, sa = _conv_bx_folded(conv_a, bn_a)
    wb, sb = _conv_bx_folded(conv_b, bn_b)
    # the sum of the two shifts, cached with the folded images it belongs to (three ATen launches per call otherwise)
    cache = getattr(conv_a, '_mas_conv_bx_dual_shift', None)
    if cache is None or cache[0] is not sa or cache[1] is not sb:
        cache = conv_a._mas_conv_bx_dual_shift = (sa, sb, (sa.double() + sb.double()).float())
    shift = cache[2]
    y = torch.empty((N, M, H, W), dtype=torch.float32, device=xa.device)
    with torch.cuda.device(xa.device):
        _lib.check(_lib.load().mas_conv_bx_fwd_dual(xa.data_ptr(), wa.data_ptr(), Ka, xb.data_ptr(), wb.data_ptr(), xb.shape[1], N, H, W, M,
                                                    shift.data_ptr(), int(relu), y.data_ptr(), _stream(xa)), "mas_conv_bx_fwd_dual")
    return y


def conv_bx(conv, x, bn=None, relu=False, residual=None):
    """relu?(bn(conv(x)) + residual) in one kernel on the bf16 matrix cores with f32 operands and results (see conv_mfma)."""
    x = x.contiguous()
    N, K, H, W = x.shape
    M = conv.out_channels
    ks, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
    wp = _conv_bx_weight(conv)
    if bn is not None and bn.num_features != M:
        raise ValueError("BatchNorm has %d features, the convolution %d output channels" % (bn.num_features, M))
    scale, shift = _bn_fold(bn) if bn is not None else (None, None)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    if residual is not None and (tuple(residual.shape) != (N, M, Ho, Wo) or residual.dtype != torch.float32 or residual.device != x.device):
        raise ValueError("residual must be float32 %s on %s, got %s %s on %s"
                         % ((N, M, Ho, Wo), x.device, residual.dtype, tuple(residual.shape), residual.device))
    res = residual.contiguous() if residual is not None else None
    y = torch.empty((N, M, Ho, Wo), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_conv_bx_fwd(x.data_ptr(), wp.data_ptr(), N, K, H, W, M, ks, s, d, _opt(scale), _opt(shift), _opt(res),
                                               int(relu), y.data_ptr(), _stream(x)), "mas_conv_bx_fwd")
    return y


class Bx3:
    """An activation tensor in presplit form (mas_bx3_split / an OUT3 epilogue): ``data`` int16 [N, ceil(C/8), 3, H*W, 8] (bf16 bits),
    ``shape`` = the (N, C, H, W) of the f32 tensor it stands for."""
    __slots__ = ("data", "shape")

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)


def bx3_split(x):
    """The presplit form of x [N,C,H,W] f32 as a pass of its own (csrc/conv_bx.hip:k_bx3_split)."""
    _need(x, "x", torch.float32)
    N, C, H, W = x.shape
    data = torch.empty((N, (C + 7) // 8, 3, H * W, 8), dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_bx3_split(x.data_ptr(), N, C, H, W, data.data_ptr(), _stream(x)), "mas_bx3_split")
    return Bx3(data, x.shape)


def conv_bx_pre(conv, x3, bn=None, relu=False, residual=None):
    """conv_bx on a presplit input (stride 1): the same products and the same bits as conv_bx(conv, x, ...) on the f32 tensor."""
    N, K, H, W = x3.shape
    M = conv.out_channels
    ks, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
    if s != 1 or K != conv.in_channels:
        raise ValueError("conv_bx_pre: stride-1 convolutions on a presplit tensor of their input channels")
    wp = _conv_bx_weight(conv)
    scale, shift = _bn_fold(bn) if bn is not None else (None, None)
    res = residual.contiguous() if residual is not None else None
    y = torch.empty((N, M, H, W), dtype=torch.float32, device=x3.data.device)
    with torch.cuda.device(y.device):
        _lib.check(_lib.load().mas_conv_bx_fwd_pre(x3.data.data_ptr(), wp.data_ptr(), N, K, H, W, M, ks, d, _opt(scale), _opt(shift), _opt(res),
                                                   int(relu), y.data_ptr(), _stream(y)), "mas_conv_bx_fwd_pre")
    return y


def stem_conv_supported(conv, x):
    """The deep stem's first convolution (3 -> C, 3x3, stride 2, padding 1) on csrc/stem.hip: fp32 NCHW, W % 8 == 0."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and conv.in_channels == 3
            and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and conv.out_channels % 16 == 0 and x.shape[3] % 8 == 0
            and x.shape[0] * (conv.out_channels // 16) <= 65535)


def stem_conv(conv, x, bn=None, relu=False):
    """relu?(bn(conv(x))) for the stem's first convolution in one kernel (inference)."""
    x = x.contiguous()
    N, _, H, W = x.shape
    M = conv.out_channels
    scale, shift = _bn_fold(bn) if bn is not None else (None, None)
    y = torch.empty((N, M, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    w = conv.weight.detach().contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_stem_conv_fwd(x.data_ptr(), w.data_ptr(), N, H, W, M, _opt(scale), _opt(shift), int(relu), y.data_ptr(),
                                                 _stream(x)), "mas_stem_conv_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# training-mode dense convolutions: forward and input gradient on the persistent stream-K kernel (csrc/conv_sk.hip), weight
# gradient on the split-K kernel (csrc/conv_wgrad.hip), all on the f32 matrix cores -- no MIOpen, no NCHW <-> NHWC copies
# ------------------------------------------------------------------------------------------------
_WGRAD_WS = {}


def _wgrad_workspace(dev, nbytes):
    """One split-K workspace per (device, stream), grown on demand (the kernels of one stream run in order, so consecutive weight
    gradients can share it; the weight gradients of a backward pass run on a side stream, _ConvTrain.backward)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WGRAD_WS[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    return ws


def conv_wgrad(x, dy, ksize, stride, dil):
    """dW [Cout,Cin,k,k] of y = conv2d(x, W, stride, padding = dil (k = 3) / 0 (k = 1), dilation) from x [N,Cin,H,W] and
    dy [N,Cout,Ho,Wo] (mas_conv_wgrad: split-K over the pixels, fixed-order reduction)."""
    _need(x, "x", torch.float32)
    _need(dy, "dy", torch.float32)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if tuple(dy.shape) != (N, Cout, Ho, Wo):
        raise ValueError("dy %s does not match x %s under stride %d" % (tuple(dy.shape), tuple(x.shape), stride))
    lib = _lib.load()
    if (ksize == 1 and stride == 1 and os.environ.get("MAS_TRAIN_BX", "auto") != "off" and Cout >= 96 and Cin >= 64
            and lib.mas_conv_wgrad_bx_supported(N, Cin, H, W, Cout)):
        # 1x1: split-bf16 kernel (csrc/conv_wgrad_bx.hip), 1.3-1.7x the f32 kernel (profiles/r04/k_bx_train_table.md); with 64 or
        # fewer output channels half of its 128 x 128 tile is padding and the f32 kernel stays ahead
        return conv_wgrad_bx(x, dy)
    if (ksize == 1 and stride == 2 and os.environ.get("MAS_TRAIN_BX", "auto") not in ("off", "r04") and Cout >= 96 and Cin >= 64
            and lib.mas_conv_wgrad_bx_supported(N, Cin, Ho, Wo, Cout)):
        # 1x1 stride 2 (`downsample`): dW only sees the even pixels of x -- gather them once (a quarter of x, one strided copy) and the
        # product is the stride-1 one on the small plane (the f32 kernel's strided K axis ran at 29 TFLOP/s: 168 us per layer)
        return conv_wgrad_bx(x[:, :, ::2, ::2].contiguous(), dy)
    w3 = os.environ.get("MAS_WGRAD3", "auto")
    if (ksize == 3 and stride == 1 and os.environ.get("MAS_TRAIN_BX", "auto") not in ("off", "r04") and w3 != "f32"
            and lib.mas_conv_wgrad_bx3_supported(N, Cin, H, W, Cout, dil)):
        # 3x3 stride 1: split-bf16 kernel with the X patch read through gfx950's transposing LDS read (csrc/conv_wgrad_bx3.hip)
        return conv_wgrad_bx3(x, dy, dil)
    nbytes = lib.mas_conv_wgrad_workspace_bytes(N, Cin, H, W, Cout, ksize, stride, dil)
    if nbytes == 0:
        raise ValueError("unsupported convolution geometry for mas_conv_wgrad")
    ws = _wgrad_workspace(x.device, nbytes)
    dw = torch.empty((Cout, Cin, ksize, ksize), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mas_conv_wgrad(x.data_ptr(), dy.data_ptr(), N, Cin, H, W, Cout, ksize, stride, dil, dw.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _stream(x)), "mas_conv_wgrad")
    return dw


def conv_wgrad_bx(x, dy):
    """dW [Cout,Cin,1,1] of a 1x1 stride-1 convolution from x [N,Cin,H,W] and dy [N,Cout,H,W] on the bf16 matrix cores with exact
    three-term splits of both f32 operands (mas_conv_wgrad_bx: split-K over the pixels, fixed-order reduction)."""
    _need(x, "x", torch.float32)
    _need(dy, "dy", torch.float32)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if tuple(dy.shape) != (N, Cout, H, W):
        raise ValueError("dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    lib = _lib.load()
    if not lib.mas_conv_wgrad_bx_supported(N, Cin, H, W, Cout):
        raise ValueError("unsupported geometry for mas_conv_wgrad_bx: x %s" % (tuple(x.shape),))
    ws = _wgrad_workspace(x.device, lib.mas_conv_wgrad_bx_workspace_bytes(Cin, Cout))
    dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mas_conv_wgrad_bx(x.data_ptr(), dy.data_ptr(), N, Cin, H, W, Cout, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _stream(x)), "mas_conv_wgrad_bx")
    return dw


def conv_wgrad_bx3(x, dy, dil=1):
    """dW [Cout,Cin,3,3] of a 3x3 stride-1 convolution (padding = dilation = 1 | 2) from x [N,Cin,H,W] and dy [N,Cout,H,W] on the bf16
    matrix cores with exact three-term splits of both f32 operands (mas_conv_wgrad_bx3: the X patch staged channel-contiguous as the
    forward kernel stages it, read with ds_read_b64_tr_b16; split-K over chunks of 4 x 16 pixels, fixed-order reduction)."""
    _need(x, "x", torch.float32)
    _need(dy, "dy", torch.float32)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if tuple(dy.shape) != (N, Cout, H, W):
        raise ValueError("dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    lib = _lib.load()
    if not lib.mas_conv_wgrad_bx3_supported(N, Cin, H, W, Cout, dil):
        raise ValueError("unsupported geometry for mas_conv_wgrad_bx3: x %s, dilation %d" % (tuple(x.shape), dil))
    ws = _wgrad_workspace(x.device, lib.mas_conv_wgrad_bx3_workspace_bytes(Cin, Cout))
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mas_conv_wgrad_bx3(x.data_ptr(), dy.data_ptr(), N, Cin, H, W, Cout, int(dil), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _stream(x)), "mas_conv_wgrad_bx3")
    return dw


_SK_WS = {}


def _sk_workspace(dev):
    """(workspace, epoch) of the stream-K convolution for the current stream of `dev`: partial-tile slots + epoch flags, zero-filled
    once; launches of one stream run in order and may share it, the epoch differs from launch to launch."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ent = _SK_WS.get(key)
    if ent is None:
        ent = _SK_WS[key] = [torch.zeros(int(_lib.load().mas_conv_sk_workspace_bytes()), dtype=torch.uint8, device=dev), 0]
    ent[1] = ent[1] % 0xfffffff0 + 1
    return ent[0], ent[1]


def conv_sk_pack(w, stride=1, dgrad=False):
    """The weight [Cout,Cin,k,k] as mas_conv_sk reads it for one role (one small launch): dgrad False / 0 forward, True / 1 input
    gradient at stride 1, 2 + sub the parity class `sub` of the input gradient of a 3x3 stride-2 convolution."""
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    lib = _lib.load()
    out = torch.empty(int(lib.mas_conv_sk_packed_elems(Cin, Cout, ks, stride, int(dgrad))), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mas_conv_sk_pack(w.data_ptr(), Cin, Cout, ks, stride, int(dgrad), out.data_ptr(), _stream(w)), "mas_conv_sk_pack")
    return out


class _PackRegistry:
    """Packed weight images of one device, re-packed together: the first use of a (weight, role) packs it alone; when a lookup
    finds a weight whose version counter or the process's parameter epoch moved (an optimizer stepped), ONE launch (mas_conv_sk_pack_multi) re-packs every
    registered image -- the optimizer updates all of them together -- instead of two small launches per convolution and step.
    Entries hold the weight WEAKLY: the active-learning loop builds a new trainer and model every round (reference train_AL.py:38),
    and a registry that kept the old weights alive would pin every dead model's convolution weights and images in HBM and re-pack
    them after every optimizer step.  An entry dies with its weight (finalizer), and dead entries never reach the job table."""

    def __init__(self, dev):
        self.dev = dev
        self.entries = {}           # key -> [weakref to the weight, image, (version, parameter epoch)]
        self.table = None           # device copy of the job records
        self.nblocks = 0
        self.njobs = 0

    @staticmethod
    def key(w, stride, dgrad):
        return (w.data_ptr(), w.untyped_storage()._cdata, tuple(w.shape), int(stride), int(dgrad))

    def _drop(self, k):
        self.entries.pop(k, None)
        self.table = None           # (its records point into freed memory)

    def get(self, w, stride, dgrad):
        import weakref
        k = self.key(w, stride, dgrad)
        e = self.entries.get(k)
        if e is not None and e[0]() is None:        # the address was recycled by a new tensor before the finalizer ran
            self._drop(k)
            e = None
        if e is None:
            img = self._pack_one(w, stride, dgrad)
            self.entries[k] = [weakref.ref(w), img, (w._version, _PARAM_EPOCH[0])]
            weakref.finalize(w, self._drop, k)
            self.table = None
            return img
        if e[2] != (w._version, _PARAM_EPOCH[0]):
            self.repack_all()
        return e[1]

    def repack_all(self):
        import ctypes
        lib = _lib.load()
        live = [(k, e, e[0]()) for k, e in list(self.entries.items())]
        for k, e, w in live:
            if w is None:
                self._drop(k)
        live = [(k, e, w) for k, e, w in live if w is not None]
        if not live:
            return
        if self.table is None:
            rec = self._job_bytes(lib)
            host = ctypes.create_string_buffer(rec * len(live))
            base = ctypes.addressof(host)
            first = 0
            for i, (k, e, w) in enumerate(live):
                n = self._fill_job(lib, base + i * rec, w, k, e[1], first)
                if n == 0:
                    raise _lib.MulActSegHipError("%s rejected the pack job %s" % (type(self).__name__, k))
                first += n
            self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.dev)
            self.nblocks = first
            self.njobs = len(live)
        with torch.cuda.device(self.dev):
            self._multi(lib, torch.cuda.current_stream(self.dev).cuda_stream)
        for k, e, w in live:
            e[2] = (w._version, _PARAM_EPOCH[0])

    # the image format of this registry: mas_conv_sk (a subclass: mas_conv_bx)
    @staticmethod
    def _pack_one(w, stride, dgrad):
        return conv_sk_pack(w, stride, dgrad)

    @staticmethod
    def _job_bytes(lib):
        return int(lib.mas_conv_sk_pack_job_bytes())

    @staticmethod
    def _fill_job(lib, rec, w, k, img, first):
        return lib.mas_conv_sk_pack_job(rec, w.data_ptr(), w.shape[1], w.shape[0], w.shape[2], k[3], int(k[4]), img.data_ptr(), first)

    def _multi(self, lib, stream):
        _lib.check(lib.mas_conv_sk_pack_multi(self.table.data_ptr(), self.njobs, self.nblocks, stream), "mas_conv_sk_pack_multi")


def conv_bx_pack(w, role=0, row_scale=None):
    """The split-bf16 weight image of mas_conv_bx_pack for one role (0 forward, 1 input gradient at stride 1); ``row_scale`` [Cout]
    (role 0): every output channel's weights multiplied by its entry before the split (a folded BatchNorm scale)."""
    _need(w, "w", torch.float32)
    w = w.contiguous()
    if row_scale is not None:
        _need(row_scale, "row_scale", torch.float32)
        if row_scale.numel() != w.shape[0] or role != 0:
            raise ValueError("row_scale: %d entries for %d output channels (role %d)" % (row_scale.numel(), w.shape[0], role))
        row_scale = row_scale.contiguous()
    M, K, kh, _ = w.shape
    lib = _lib.load()
    nbytes = lib.mas_conv_bx_packed_bytes(kh, K, M, int(role))
    if nbytes <= 0:
        raise ValueError("mas_conv_bx_pack does not take a %dx%d weight %s" % (kh, kh, tuple(w.shape)))
    wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mas_conv_bx_pack(w.data_ptr(), _opt(row_scale), M, K, kh, int(role), wp.data_ptr(), _stream(w)), "mas_conv_bx_pack")
    return wp


class _BxPackRegistry(_PackRegistry):
    """The same bookkeeping for the split-bf16 images of csrc/conv_bx.hip (key: stride is always 1, `dgrad` is the role)."""

    @staticmethod
    def _pack_one(w, stride, dgrad):
        return conv_bx_pack(w, int(dgrad))

    @staticmethod
    def _job_bytes(lib):
        return int(lib.mas_conv_bx_pack_job_bytes())

    @staticmethod
    def _fill_job(lib, rec, w, k, img, first):
        return lib.mas_conv_bx_pack_job(rec, w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], int(k[4]), img.data_ptr(), first)

    def _multi(self, lib, stream):
        _lib.check(lib.mas_conv_bx_pack_multi(self.table.data_ptr(), self.njobs, self.nblocks, stream), "mas_conv_bx_pack_multi")


_PACKS = {}
_BX_PACKS = {}


def bx_packed_weight(w, role=0):
    """The mas_conv_bx image of weight `w` for one role, kept up to date across optimizer steps (one re-pack launch per step for
    all registered weights: _BxPackRegistry)."""
    reg = _BX_PACKS.get(w.device)
    if reg is None:
        reg = _BX_PACKS[w.device] = _BxPackRegistry(w.device)
    return reg.get(w, 1, int(role))


_BX_WS = {}


def _bx_workspace(dev, nbytes):
    """Per-(device, stream) scratch for the partial tiles of a split-K mas_conv_bx_train launch (grown on demand, never shrunk; the
    launches of one stream run in order, so they can share it)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _BX_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _BX_WS[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
    return ws


def conv_bx_train_plan(x_shape, w_shape, dil=1, dgrad=False):
    """(ksplit, tile_w, workgroups) mas_conv_bx_train would choose for this product (mas_conv_bx_train_plan)."""
    import ctypes
    Cout, Cin, ks, _ = w_shape
    K, M = (Cout, Cin) if dgrad else (Cin, Cout)
    N, _, H, W = x_shape
    out = (ctypes.c_int * 3)()
    _lib.check(_lib.load().mas_conv_bx_train_plan(N, K, H, W, M, ks, dil, out), "mas_conv_bx_train_plan")
    return int(out[0]), int(out[1]), int(out[2])


def conv_bx_s2_raw(x, w, packed=None):
    """y = conv2d(x, w, stride 2[, padding 1]) for a 1x1 or 3x3 weight on the stride-2 form of csrc/conv_bx.hip (bare product of a
    training step).  packed: the image of role 0 (1x1) / BX_ROLE_S2 (3x3)."""
    _need(x, "x", torch.float32)
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    N, Cx, H, W = x.shape
    if ks not in (1, 3) or Cx != Cin:
        raise ValueError("conv_bx_s2_raw: a 1x1 or 3x3 weight on its input channels")
    y = torch.empty((N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    if packed is None:
        packed = conv_bx_pack(w, 0 if ks == 1 else BX_ROLE_S2)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mas_conv_bx_fwd(x.data_ptr(), packed.data_ptr(), N, Cin, H, W, Cout, ks, 2, 1, None, None, None, 0, y.data_ptr(),
                                               _stream(x)), "mas_conv_bx_fwd")
    return y


def conv_bx_raw(x, w, dil=1, dgrad=False, residual=None, packed=None, ksplit=0, tile_w=0, stats=False):
    """The bare stride-1 product of a training step on csrc/conv_bx.hip: dgrad False: y = conv2d(x, w, padding = dil (k 3) / 0 (k 1),
    dilation); dgrad True: x is dY [N,Cout,H,W] and the result dX [N,Cin,H,W] (+ residual: the gradient of x's other consumer).
    ksplit / tile_w: 0 = the library's work-splitting plan (mas_conv_bx_train_plan); explicit values for sweeps and tests.
    stats=True (forward, no residual): returns (y, partials) -- partials [Cout, slots, 2] float64, the BatchNorm partial sums
    (sum y, sum y^2 over disjoint pixel sets) formed in the kernel's epilogue (ksplit 1) or by the reduction pass of a split-K plan, or
    None where neither can (split K on a plane of odd size).
    Operands must be finite and within 2^-100 < |v| < 2^127 (csrc/bx_split.h): an Inf operand yields NaN where the f32 pipe would
    propagate Inf (h = Inf, m = Inf - Inf), NaN stays NaN, values below 2^-100 lose their third term (relative error <= 2^-16)."""
    _need(x, "x", torch.float32)
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    N, Cx, H, W = x.shape
    K, M = (Cout, Cin) if dgrad else (Cin, Cout)
    if Cx != K:
        raise ValueError("input has %d channels, weight %s (dgrad=%s)" % (Cx, tuple(w.shape), dgrad))
    y = torch.empty((N, M, H, W), dtype=torch.float32, device=x.device)
    if residual is not None:
        _need(residual, "residual", torch.float32)
        if residual.shape != y.shape:
            raise ValueError("residual %s does not match the output %s" % (tuple(residual.shape), tuple(y.shape)))
    if packed is None:
        packed = conv_bx_pack(w, int(dgrad))
    lib = _lib.load()
    if not ksplit or not tile_w:
        plan = conv_bx_train_plan(x.shape, w.shape, dil, dgrad)
        ksplit = ksplit or plan[0]
        tile_w = tile_w or plan[1]
        cap = os.environ.get("MAS_BX_KSPLIT_DGRAD" if dgrad else "MAS_BX_KSPLIT_FWD")       # (A/B: cap the plan's K split)
        if cap:
            ksplit = max(1, min(ksplit, int(cap)))
    if ksplit > 1 and (y.numel() % 4 != 0):
        ksplit = 1                                   # (the reduction pass walks 16-byte groups)
    ws = part = None
    with torch.cuda.device(x.device):
        if ksplit > 1:
            ws = _bx_workspace(x.device, int(lib.mas_conv_bx_train_workspace_bytes(N, M, H, W, ksplit)))
        if stats and not dgrad and residual is None:
            slots = int(lib.mas_conv_bx_train_stat_slots(N, H, W, M, ks, dil, int(ksplit), int(tile_w)))
            if slots > 0:
                part = torch.empty((M, slots, 2), dtype=torch.float64, device=x.device)
        _lib.check(lib.mas_conv_bx_train(x.data_ptr(), packed.data_ptr(), N, K, H, W, M, ks, dil, _opt(residual), y.data_ptr(), int(ksplit),
                                         int(tile_w), _opt(ws), ws.numel() if ws is not None else 0, _opt(part), _stream(x)), "mas_conv_bx_train")
    return (y, part) if stats else y


def conv_bx_train_ok(x_shape, w_shape, stride, dil, dgrad):
    """Does the split-bf16 kernel take this product of a training step?  Stride 1 only.  MAS_TRAIN_BX = auto (default): every
    supported stride-1 product -- launches with fewer workgroups than the chip has slots (the 48 x 48 planes) split their K chunks
    over several workgroups and use 16 x 16 pixel tiles (mas_conv_bx_train_plan; profiles/r05/k_bx_train_table.md); r04: round 4's
    rule (1x1 everywhere, 3x3 on planes of at least 96 x 96 or with at least 320 tiles -- the others on the persistent stream-K
    kernel), for A/B runs; off: never."""
    mode = os.environ.get("MAS_TRAIN_BX", "auto")
    Cout, Cin, ks, _ = w_shape
    N, _, H, W = x_shape
    if mode not in ("off", "r04") and stride == 2 and not dgrad and (ks == 1 or (ks == 3 and dil == 1 and os.environ.get("MAS_BX_S2K3", "on") != "off")):
        # the 1x1 stride-2 `downsample` convolutions (resnet.py:215-223) and the 3x3 stride-2 conv2 of layer2.0 / layer3.0 (:140-150):
        # the stride-2 form of the forward kernel (planes with even H and W % 8 == 0 -- the 768 crop; the 769 crop's odd planes stay
        # on the stream-K kernel)
        if not (_lib.load().mas_conv_bx_supported(ks, 2, 1, Cin, Cout, H, W) and x_shape[0] > 0):
            return False
        # (the strided 3x3 has no work splitting: below 256 workgroups -- layer3.0.conv2 on the 48 x 48 planes of the training crop,
        #  144 of them with 72 chunks each -- the persistent stream-K kernel is faster: 114 against 142 us, tools/s2k3_probe.py)
        bm = 128 if Cout % 128 == 0 else 64
        wgs = (N * (H // 2) * (W // 2) + (16384 // bm) - 1) // (16384 // bm) * ((Cout + bm - 1) // bm)      # (pixel tile: 128 / 256)
        return ks == 1 or wgs >= 256
    if mode == "off" or stride != 1:
        return False
    K, M = (Cout, Cin) if dgrad else (Cin, Cout)
    if not _lib.load().mas_conv_bx_supported(ks, 1, dil, K, M, H, W):
        return False
    if mode != "r04" or ks == 1 or H * W >= 96 * 96:
        return True
    return N * ((H + 7) // 8) * ((W + 31) // 32) * ((M + 63) // 64) >= 320


def packed_weight(w, stride=1, dgrad=False):
    """The mas_conv_sk image of weight `w` for one role, kept up to date across optimizer steps (see _PackRegistry)."""
    reg = _PACKS.get(w.device)
    if reg is None:
        reg = _PACKS[w.device] = _PackRegistry(w.device)
    return reg.get(w, stride, dgrad)


_SK_DEFAULT_FLAGS = [0]         # wrapper default of the per-call mas_sk_opts.flags (conv_sk_set_mode: A/B runs of the LDS-DMA ring)


def sk_split_default():
    """Does a conv_sk call without explicit flags split tiles over workgroups (the stream-K hand-off)?  MAS_SK_SPLIT = auto
    (default): yes in a single-GPU process, NO under torch.distributed with more than one rank -- the hand-off assumes that the
    workgroups of a launch become resident together, which a neighbour holding CUs for seconds (an RCCL kernel of a wedged peer,
    another tenant) can break; a finisher that gives up poisons its tile and the step aborts (trainer/base.py:raise_stream_k),
    and one rank aborting while the others sit in the gradient all-reduce is a hang.  The whole-tile plan (SK_NOSPLIT) has no
    hand-off and is exact; it costs the load balance of the few layers that still run on k_conv_sk.
    on / off force either."""
    mode = os.environ.get("MAS_SK_SPLIT", "auto")
    if mode in ("on", "1"):
        return True
    if mode in ("off", "0"):
        return False
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _sk_opts(flags=None, spin_limit=0, stamps=None):
    """mas_sk_opts for one call (NULL when everything is the default)."""
    import ctypes
    if flags is None:
        flags = _SK_DEFAULT_FLAGS[0] | (0 if sk_split_default() else _lib.SK_NOSPLIT)
    flags = int(flags)
    if not flags and not spin_limit and stamps is None:
        return None, flags
    o = _lib.SkOpts(flags, int(spin_limit), stamps.data_ptr() if stamps is not None else None)
    return ctypes.byref(o), flags


def conv_sk(x, w, stride=1, dil=1, dgrad=False, scale=None, shift=None, residual=None, relu=False, packed=None, stats=False,
            flags=None, spin_limit=0, stamps=None):
    """Training-mode dense convolution on the persistent stream-K MFMA kernel (mas_conv_sk), weight `w` [Cout,Cin,k,k] as PyTorch
    stores it (``packed``: its conv_sk_pack image for this role, when the caller keeps one).  dgrad=False: y = conv2d(x, w, stride, padding = dil (k 3) / 0 (k 1), dilation); dgrad=True: x is dY [N,Cout,H,W] and
    the result is dX [N,Cin,H,W] of the stride-1 convolution.  Optional epilogue y*scale[m] + shift[m] + residual, ReLU.
    ``flags`` (_lib.SK_DMA | _lib.SK_NOSPLIT), ``spin_limit``, ``stamps``: the per-call mas_sk_opts."""
    _need(x, "x", torch.float32)
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    N, Cx, H, W = x.shape
    if Cx != (Cout if dgrad else Cin):
        raise ValueError("input has %d channels, weight %s (dgrad=%s)" % (Cx, tuple(w.shape), dgrad))
    M = Cin if dgrad else Cout
    Ho, Wo = (H, W) if dgrad else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    y = torch.empty((N, M, Ho, Wo), dtype=torch.float32, device=x.device)
    if residual is not None:
        _need(residual, "residual", torch.float32)
        if residual.shape != y.shape:
            raise ValueError("residual %s does not match the output %s" % (tuple(residual.shape), tuple(y.shape)))
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None:
            _need(t, name, torch.float32)
            if t.numel() != M:
                raise ValueError("%s must have %d entries" % (name, M))
    if packed is None:
        packed = conv_sk_pack(w, stride, dgrad)
    ws, epoch = _sk_workspace(x.device)
    lib = _lib.load()
    opts, flags = _sk_opts(flags, spin_limit, stamps)
    if stats:
        # forward without epilogue + the BatchNorm partial sums of y from the epilogue of every tile: (y, partials [Cout, slots, 2] f64)
        if dgrad or scale is not None or residual is not None or relu:
            raise ValueError("stats=True: bare forward product only")
        slots = int(lib.mas_conv_sk_stats_slots(N, Cin, H, W, Cout, ks, stride, dil, flags))
        if slots <= 0:
            raise ValueError("unsupported geometry for mas_conv_sk_stats")
        part = torch.empty((Cout, slots, 2), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.mas_conv_sk_stats(x.data_ptr(), packed.data_ptr(), N, Cin, H, W, Cout, ks, stride, dil, y.data_ptr(), part.data_ptr(),
                                             ws.data_ptr(), ws.numel(), epoch, opts, _stream(x)), "mas_conv_sk_stats")
        return y, part
    with torch.cuda.device(x.device):
        _lib.check(lib.mas_conv_sk(x.data_ptr(), packed.data_ptr(), N, Cin, H, W, Cout, ks, stride, dil, int(dgrad), _opt(scale), _opt(shift),
                                   _opt(residual), int(relu), y.data_ptr(), ws.data_ptr(), ws.numel(), epoch, opts, _stream(x)),
                   "mas_conv_sk")
    return y


def conv_sk_dgrad_s2(dy, w, H, W, packed=None, flags=None, spin_limit=0):
    """dX [N,Cin,H,W] of y = conv2d(x, w, stride 2, padding 1) for a 3x3 weight `w` [Cout,Cin,3,3] from dY [N,Cout,(H-1)//2+1,
    (W-1)//2+1]: four launches of the stream-K kernel, one per parity class of the dX pixels (mas_conv_sk_dgrad_s2) -- each a
    stride-1 product over the dY plane with 1 / 2 / 2 / 4 taps, together the exact FLOPs of the gradient (no zero insertion).
    ``packed``: the four class images (conv_sk_pack(w, 2, 2 + sub))."""
    _need(dy, "dy", torch.float32)
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    N = dy.shape[0]
    if ks != 3 or tuple(dy.shape) != (N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise ValueError("dy %s does not belong to a 3x3 stride-2 convolution of a %dx%d plane with weight %s" % (tuple(dy.shape), H, W, tuple(w.shape)))
    dx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    opts, _ = _sk_opts(flags, spin_limit)
    with torch.cuda.device(dy.device):
        for sub in range(4):
            img = packed[sub] if packed is not None else conv_sk_pack(w, 2, 2 + sub)
            ws, epoch = _sk_workspace(dy.device)
            _lib.check(lib.mas_conv_sk_dgrad_s2(dy.data_ptr(), img.data_ptr(), N, Cin, H, W, Cout, sub, None, None, None, 0, dx.data_ptr(),
                                                ws.data_ptr(), ws.numel(), epoch, opts, _stream(dy)), "mas_conv_sk_dgrad_s2")
    return dx


def conv_sk_dgrad_1x1s2(dy, w, H, W, packed=None):
    """dX [N,Cin,H,W] of y = conv2d(x, w, stride 2) for a 1x1 weight `w` [Cout,Cin,1,1] from dY [N,Cout,(H-1)//2+1,(W-1)//2+1]: the
    gradient lives on the even positions only.  ONE launch of the stream-K kernel: the stride-1 product over the dY plane with the
    strided store of the stride-2 parity classes (class 0 of mas_conv_sk_dgrad_s2 is exactly a one-tap product written to the pixels
    (2i, 2j); it takes the 1x1 weight's input-gradient image) into a zero-filled dX.  (Until round 4: the product into a temporary,
    then zeros_like + a strided ATen copy.)"""
    _need(dy, "dy", torch.float32)
    _need(w, "w", torch.float32)
    Cout, Cin, ks, _ = w.shape
    N = dy.shape[0]
    if ks != 1 or tuple(dy.shape) != (N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise ValueError("dy %s does not belong to a 1x1 stride-2 convolution of a %dx%d plane with weight %s" % (tuple(dy.shape), H, W, tuple(w.shape)))
    dx = torch.zeros((N, Cin, H, W), dtype=torch.float32, device=dy.device)
    img = packed if packed is not None else conv_sk_pack(w, 1, True)
    lib = _lib.load()
    opts, _ = _sk_opts()
    ws, epoch = _sk_workspace(dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.mas_conv_sk_dgrad_s2(dy.data_ptr(), img.data_ptr(), N, Cin, H, W, Cout, 0, None, None, None, 0, dx.data_ptr(),
                                            ws.data_ptr(), ws.numel(), epoch, opts, _stream(dy)), "mas_conv_sk_dgrad_s2")
    return dx


def conv_sk_set_mode(dma):
    """Default chunk staging of this wrapper's conv_sk calls (the library itself keeps no mode: the flag travels with every call in
    mas_sk_opts): False = register-staged (default), True = LDS-DMA ring; returns the previous default.  For A/B runs and tests."""
    old = bool(_SK_DEFAULT_FLAGS[0] & _lib.SK_DMA)
    _SK_DEFAULT_FLAGS[0] = (_SK_DEFAULT_FLAGS[0] & ~_lib.SK_DMA) | (_lib.SK_DMA if dma else 0)
    return old


def conv_sk_error(dev=None):
    """Non-zero when a stream-K launch on the current stream's workspace gave up waiting for another workgroup (synchronises)."""
    import ctypes
    dev = torch.device('cuda', torch.cuda.current_device()) if dev is None else dev
    ent = _SK_WS.get((dev, torch.cuda.current_stream(dev).cuda_stream))
    if ent is None:
        return 0
    torch.cuda.synchronize(dev)
    out = ctypes.c_uint(0)
    _lib.check(_lib.load().mas_conv_sk_error(ent[0].data_ptr(), ctypes.byref(out)), "mas_conv_sk_error")
    return int(out.value)


def _cuda_dev(dev):
    dev = torch.device('cuda', torch.cuda.current_device()) if dev is None else torch.device(dev)
    return torch.device('cuda', torch.cuda.current_device()) if dev.index is None else dev


def _sk_error_views(dev):
    off = int(_lib.load().mas_conv_sk_workspace_bytes()) - 4096 + 512 * 4       # the word behind the 512 epoch flags
    return [ent[0][off:off + 4] for (d, _), ent in _SK_WS.items() if d == dev]


def conv_sk_error_words(dev=None):
    """The error words of every stream-K workspace of `dev` as ONE device tensor [n] int32 (gathered on the current stream, no
    synchronisation), or None when no stream-K launch has run there yet.  The words are sticky (a finisher that gives up ORs a bit
    in; nothing but conv_sk_clear_error clears it), so a caller may look at them late: the trainers ship them to the host with the
    loss of the NEXT step (trainer/active_joint_multi.py:_HostProbe) and raise StreamKGaveUp."""
    views = _sk_error_views(_cuda_dev(dev))
    return torch.cat(views).view(torch.int32) if views else None


def conv_sk_clear_error(dev=None):
    """Zero the error words of `dev` (after the caller has dealt with a reported give-up)."""
    for v in _sk_error_views(_cuda_dev(dev)):
        v.zero_()


class StreamKGaveUp(RuntimeError):
    """A stream-K convolution gave up waiting for another workgroup of its launch: the step's activations / gradients are poisoned
    (NaN).  Seen only when the GPU is shared with something that keeps CUs busy for seconds."""


_SIDE_STREAMS = {}      # device -> the side stream of MAS_WGRAD_STREAM=side


def _side_stream(dev):
    """The stream the weight gradients run on under MAS_WGRAD_STREAM=side (dW is a leaf of the backward graph; not the default:
    see _ConvTrain.backward)."""
    st = _SIDE_STREAMS.get(dev)
    if st is None:
        st = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


_BRANCH_STREAMS = {}


def branch_stream(dev, k):
    """Stream number k for an independent BRANCH of the network in a training step (the downsample path of a Bottleneck beside its
    conv1-conv2-conv3 chain, the ASPP branches, the decoder's low-level projection): most layers of the 48 x 48 / 96 x 96 planes
    launch fewer workgroups than the chip has slots, so two independent layers side by side finish sooner than one after the other.
    autograd replays every node's backward on the stream of its forward and orders the streams itself, so the backward pass of a
    branch overlaps the same way.  MAS_BRANCH_STREAMS=on; the DEFAULT IS OFF (None): the step gains 0.2 ms (23.95 -> 23.74 at the 768
    crop, same bits), but the autograd engine record_stream()s every gradient that crosses streams, and blocks freed with a recorded
    stream use cannot serve the next step's allocations while the host runs ahead of the device -- 1.7 hipMalloc calls per step in
    steady state, reserved memory 9.6 -> 16.6 GB (profiles/r06/wgrad_stream_ab.md).  A 1 % gain is not worth an allocator that
    never settles."""
    if os.environ.get("MAS_BRANCH_STREAMS", "off") != "on":
        return None
    key = (dev, int(k))
    st = _BRANCH_STREAMS.get(key)
    if st is None:
        st = _BRANCH_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def run_on_branch(stream, fn, *tensors):
    """fn(*tensors) on `stream` (which first waits for the current stream: the inputs are ready there); the caller joins with
    join_branch(stream, outputs) before the outputs are used on the current stream."""
    cur = torch.cuda.current_stream(tensors[0].device)
    stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        out = fn(*tensors)
    # (no Tensor.record_stream -- see _ConvTrain.backward: the inputs are activations that live on until the backward pass, i.e. past
    #  join_branch, after which every free on the current stream is ordered behind the branch's reads; the outputs come from the branch
    #  stream's pool and their blocks are reused by that stream only after it has waited for the current stream again)
    return out


def join_branch(stream, *outputs):
    torch.cuda.current_stream(outputs[0].device).wait_stream(stream)


_JOIN_QUEUED = set()
_KEEP_UNTIL_JOIN = {}          # device -> [(x, dy), ...] of the weight gradients queued on the side stream in this backward pass


def _async_wgrad_ok(w):
    """The weight gradient may finish on the side stream after its autograd node has returned only if NOTHING reads it before the
    end-of-backward join: w.grad is None (AccumulateGrad then only stores the tensor; with a gradient buffer in place it adds on the
    main stream), no gradient hooks on w, and no data-parallel reducer (DistributedDataParallel copies / all-reduces a gradient the
    moment its AccumulateGrad node has run)."""
    if w.grad is not None or getattr(w, '_post_accumulate_grad_hooks', None) or getattr(w, '_backward_hooks', None):
        return False
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized())      # (any process group: a reducer hooks the AccumulateGrad nodes, not the tensors)


def _join_side_after_backward(dev):
    """Queue ONE callback per backward pass (torch.autograd's engine runs it when the pass has finished, before .backward() returns):
    the main stream waits for the side stream there."""
    if dev in _JOIN_QUEUED:
        return
    _JOIN_QUEUED.add(dev)

    def join():
        _JOIN_QUEUED.discard(dev)
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
        _KEEP_UNTIL_JOIN.pop(dev, None)             # (only now may the operands of the weight gradients be freed: see _ConvTrain.backward)
    torch.autograd.Variable._execution_engine.queue_callback(join)


def _aten_pad(ks, dil):
    return (dil, dil) if ks == 3 else (0, 0)


class _ConvTrain(torch.autograd.Function):
    """y = conv2d(x, w) with autograd.  ``own`` = (forward, input gradient, weight gradient) on this package's f32-MFMA kernels:
    forward and input gradient on the stream-K kernel k_conv_sk (it reads the weight as PyTorch stores it; the input gradient
    is the same kernel with the weight's channel axes swapped and the taps mirrored; stride 1), weight gradient on k_wgrad; a
    False entry takes MIOpen through ATen for that product.
    stats: also returns the BatchNorm partial sums of y (formed in the forward kernel's epilogue).  fork: also returns an alias of x
    for the OTHER consumer of x (the residual branch of a Bottleneck: models/segmentation/backbone/resnet.py:143-160): the gradient
    that arrives through the alias is added in the epilogue of the input-gradient kernel (its `residual` operand) instead of by a
    separate pass over both gradients (autograd's accumulation: 16 add kernels over ~1.1 GB per step)."""

    @staticmethod
    def forward(ctx, x, w, stride, dil, own, stats=False, fork=False):
        if _JOIN_QUEUED:                    # a backward pass that raised before its end-of-pass callback ran: join the side stream now
            for dev in list(_JOIN_QUEUED):
                _JOIN_QUEUED.discard(dev)
                torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
                _KEEP_UNTIL_JOIN.pop(dev, None)
        x = x.contiguous()
        ks = w.shape[2]
        part = None
        with torch.no_grad():
            if own[0] and conv_bx_train_ok(x.shape, w.shape, stride, dil, False):
                # split-bf16 kernel (csrc/conv_bx.hip); the BatchNorm partial sums then come from the separate reduction pass
                if stride == 2:
                    y = conv_bx_s2_raw(x, w, packed=bx_packed_weight(w, 0 if w.shape[2] == 1 else BX_ROLE_S2))
                elif stats:
                    # (the BatchNorm partial sums from the kernel's epilogue / the split-K reduction pass)
                    y, part = conv_bx_raw(x, w, dil, packed=bx_packed_weight(w, 0), stats=True)
                else:
                    y = conv_bx_raw(x, w, dil, packed=bx_packed_weight(w, 0))
            elif own[0] and stats:
                y, part = conv_sk(x, w, stride, dil, packed=packed_weight(w, stride, False), stats=True)
            elif own[0]:
                y = conv_sk(x, w, stride, dil, packed=packed_weight(w, stride, False))
            else:
                y = torch.nn.functional.conv2d(x, w, None, stride, _aten_pad(ks, dil), dil)
        ctx.save_for_backward(x, w)
        ctx.geom = (ks, stride, dil, own)
        ctx.fork = bool(fork)
        if not stats and not fork:
            return y
        ctx.set_materialize_grads(False)            # (no zero-filled "gradient" of the partial sums in the backward)
        outs = [y]
        if stats:
            if part is not None:
                ctx.mark_non_differentiable(part)
            outs.append(part)
        if fork:
            outs.append(x.view_as(x))
        return tuple(outs)

    @staticmethod
    def backward(ctx, dy, *rest):
        x, w = ctx.saved_tensors
        ks, stride, dil, own = ctx.geom
        g_other = rest[-1] if (ctx.fork and rest) else None     # gradient of the alias = of the other consumer of x
        if dy is None:
            return g_other, None, None, None, None, None, None
        dy = dy.contiguous()
        dx = dw = None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        side = None
        if need_dw and own[2]:
            main = torch.cuda.current_stream(x.device)
            # MAS_WGRAD_STREAM: main = one stream; side = dW on a second stream, joined behind the input gradient of the SAME layer
            # (round 3, persistent one-workgroup-per-CU kernels: 34.1 vs 32.4 ms; round 6, tile kernels: 27.2 vs 25.0 ms -- it loses
            # either way); async (the default where _async_wgrad_ok) = the second stream joined once per backward pass.
            mode = os.environ.get("MAS_WGRAD_STREAM", "async")
            if mode == "async" and not _async_wgrad_ok(w):
                mode = "main"
            if mode == "async" or (mode == "side" and need_dx):
                side = _side_stream(x.device)
                side.wait_stream(main)                      # dy (and x) are ready on the main stream
                with torch.cuda.stream(side):
                    dw = conv_wgrad(x, dy, ks, stride, dil)
                # Lifetimes across the two streams WITHOUT Tensor.record_stream: a block freed with a recorded stream use is held back
                # until an event has completed, and when the host runs ahead of the device (a step queues in 17 ms, runs in 23) those
                # held-back blocks cannot serve the next step's allocations -- the caching allocator then calls hipMalloc in steady
                # state (measured: 12-15 calls per step, reserved memory growing with the run-ahead depth, and on some boxes a 769 step
                # of 38 ms instead of 26).  Instead: x and dy are kept ALIVE until the join (a reference in _KEEP_UNTIL_JOIN), so their
                # blocks return to the main stream's pool only after the main stream has waited for the side stream; dW is allocated
                # from the side stream's pool, read by the optimizer behind the join, and its block is reused by the side stream only
                # after that stream has waited for the main stream again (the wait above, one step later).
                if mode == "async":
                    # dW is a LEAF of the backward graph: nothing downstream of this node reads it before the optimizer does.  The side
                    # stream is joined ONCE, when the whole backward pass has been queued (engine callback), so the weight gradients
                    # of all layers drain beside the chain of input gradients and BatchNorm backward passes -- most of whose launches
                    # leave CUs idle (144 workgroups of a 48 x 48 layer on 256 CUs) -- instead of being waited for layer by layer:
                    # 25.0 -> 23.8 ms per step at the 768 crop, same bits (profiles/r06/wgrad_stream_ab.md).  ("side", the per-layer
                    # join: 27.2 ms -- the join makes every layer wait for the slower of its two products.)
                    _KEEP_UNTIL_JOIN.setdefault(x.device, []).append((x, dy))
                    _join_side_after_backward(x.device)
                    side = None
            else:
                dw = conv_wgrad(x, dy, ks, stride, dil)
        if need_dx:
            if own[1] and stride == 1 and conv_bx_train_ok(dy.shape, w.shape, 1, dil, True):
                fuse = g_other is not None and g_other.shape == x.shape and g_other.dtype == torch.float32
                dx = conv_bx_raw(dy, w, dil, dgrad=True, residual=g_other.contiguous() if fuse else None, packed=bx_packed_weight(w, 1))
                if fuse:
                    g_other = None
            elif own[1] and stride == 1:
                if g_other is not None and g_other.shape == x.shape and g_other.dtype == torch.float32:
                    dx = conv_sk(dy, w, 1, dil, dgrad=True, packed=packed_weight(w, 1, True), residual=g_other.contiguous())
                    g_other = None
                else:
                    dx = conv_sk(dy, w, 1, dil, dgrad=True, packed=packed_weight(w, 1, True))
            elif own[1] and ks == 1:
                # 1x1, stride 2: the input gradient lives on the even positions only -- the stride-1 product over the small plane,
                # stored with stride 2 into a zero-filled tensor by the kernel itself
                if stride == 2:
                    dx = conv_sk_dgrad_1x1s2(dy, w, x.shape[2], x.shape[3], packed=packed_weight(w, 1, True))
                else:
                    dx = torch.zeros_like(x)
                    dx[:, :, ::stride, ::stride] = conv_sk(dy, w, 1, 1, dgrad=True, packed=packed_weight(w, 1, True))
            elif own[1] and ks == 3 and stride == 2 and dil == 1:
                dx = conv_sk_dgrad_s2(dy, w, x.shape[2], x.shape[3], packed=[packed_weight(w, 2, 2 + sub) for sub in range(4)])
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, (stride, stride), _aten_pad(ks, dil), (dil, dil), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if need_dw and not own[2]:
            dw = torch.ops.aten.convolution_backward(dy, x, w, None, (stride, stride), _aten_pad(ks, dil), (dil, dil), False, (0, 0), 1,
                                                     (False, True, False))[1]
        if side is not None:
            main.wait_stream(side)                          # dW joins the main stream behind the input gradient (x, dy: alive until here)
        if g_other is not None:
            dx = g_other if dx is None else dx + g_other
        return dx, dw, None, None, None, None, None


def conv_wgrad_supported(conv, x):
    """mas_conv_wgrad takes every dense 1x1 / 3x3 convolution of the network: any channel counts and plane sizes, stride 1 / 2,
    padding = dilation (3x3) / 0 (1x1), dilation 1 / 2 / 4 at stride 1."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and conv.bias is None):
        return False
    k, s, d, pd = conv.kernel_size, conv.stride, conv.dilation, conv.padding
    if k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or pd[0] != pd[1] or k[0] not in (1, 3) or s[0] not in (1, 2):
        return False
    if k[0] == 3 and (pd[0] != d[0] or d[0] not in (1, 2, 4) or (s[0] == 2 and d[0] != 1)):
        return False
    if k[0] == 1 and (pd[0] != 0 or d[0] != 1):
        return False
    return x.shape[1] == conv.in_channels and max(conv.in_channels, conv.out_channels) * x.shape[2] * x.shape[3] < 2 ** 31


def conv_train_plan(conv, x):
    """(forward, input gradient, weight gradient) -> True where this package's kernel runs the product, or None when the layer
    is outside all three (then the caller keeps the nn.Module call).  MAS_TRAIN_CONV = own (default): every product the kernels
    support -- all three of every dense convolution of the network; miopen: none; auto: the weight
    gradient everywhere, forward / input gradient only on planes of >= 192 x 192 / 384 x 384 pixels (MIOpen's Tensile GEMMs are
    still ahead on the 1x1 layers of the small planes: profiles/r03/b_conv_train_table_streamk.md; ~0.4 ms per step)."""
    mode = os.environ.get("MAS_TRAIN_CONV", "own")
    if mode == "miopen" or not conv_wgrad_supported(conv, x):
        return None
    if _reg_step_hook is None and _PARAM_EPOCH[0] == 0:     # (no optimizer hook and nobody invalidates by hand: see the import above)
        return None
    hw = x.shape[2] * x.shape[3]
    fwd_ok = conv.dilation[0] in (1, 2, 4) and hw >= 64
    dgrad_ok = fwd_ok and (conv.stride[0] == 1 or conv.kernel_size[0] == 1 or conv.dilation[0] == 1)
    wgrad_ok = conv.in_channels >= 8
    if mode == "own":
        return (fwd_ok, dgrad_ok, True)
    return (fwd_ok and hw >= 192 * 192, dgrad_ok and hw >= 384 * 384, wgrad_ok)


def conv_train(conv, x, own=(True, True, True), stats=False, fork=False):
    """conv(x) with autograd on the package's kernels (see _ConvTrain).  stats=True: (y, partials) -- the BatchNorm partial sums
    of y from the epilogue of the forward kernel (None when the forward product is not on mas_conv_sk), for bn_act(partials=).
    fork=True: one more result, an alias of x to hand to the other consumer of x (see _ConvTrain)."""
    own = tuple(bool(v) for v in own)
    stats = bool(stats and own[0])
    out = _ConvTrain.apply(x, conv.weight, conv.stride[0], conv.dilation[0], own, stats, bool(fork))
    if not stats and not fork:
        return out
    out = list(out)
    if not stats:
        out.insert(1, None)
    return tuple(out) if fork else (out[0], out[1])
