"""ctypes wrapper of oracle/libexact.so (the plain-C restatement in the normative arithmetic).

ORACLE / TEST INFRASTRUCTURE ONLY -- see oracle/exact.c.  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libexact.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libexact.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def inv_temperature(T):
    """invT exactly as the product computes it: float32(1 / float32(T))."""
    return np.float32(1.0) / np.float32(T)


def class_prob_sum(z, invT):
    z = _c(z, np.float32)
    B, C, H, W = z.shape
    out = np.zeros((B, C), dtype=np.uint64)
    lib().exact_class_prob_sum(_p(z), B, C, H, W, ctypes.c_float(invT), _p(out))
    return out


def class_weight(prob_sum, HW, batch_of, n_batches, coeff):
    prob_sum = _c(prob_sum, np.uint64)
    n_img, C = prob_sum.shape
    batch_of = _c(batch_of, np.int32)
    cum = np.zeros(C, dtype=np.float64)
    w = np.zeros(C, dtype=np.float32)
    lib().exact_class_weight(_p(prob_sum), n_img, C, ctypes.c_int64(HW), _p(batch_of), n_batches,
                             ctypes.c_double(coeff), _p(cum), _p(w))
    return cum, w


def bvsb_region_accum(z, spx, cls_w, S, invT):
    z = _c(z, np.float32)
    spx = _c(spx, np.int64)
    B, C, H, W = z.shape
    ssum = np.zeros((B, S), dtype=np.uint64)
    hist = np.zeros((B, S, C), dtype=np.uint32)
    w = None if cls_w is None else _c(cls_w, np.float32)
    lib().exact_bvsb_region_accum(_p(z), _p(spx), None if w is None else _p(w), B, C, H, W, S,
                                  ctypes.c_float(invT), _p(ssum), _p(hist))
    return ssum, hist


def region_finalize(score_sum, hist, ban_class):
    score_sum = _c(score_sum, np.uint64)
    hist = _c(hist, np.uint32)
    C = hist.shape[-1]
    n = score_sum.size
    score = np.zeros(score_sum.shape, dtype=np.float32)
    dom = np.zeros(score_sum.shape, dtype=np.int32)
    cnt = np.zeros(score_sum.shape, dtype=np.uint32)
    lib().exact_region_finalize(_p(score_sum), _p(hist), ctypes.c_int64(n), C, ban_class, _p(score), _p(dom), _p(cnt))
    return score, dom, cnt


def expf(x):
    x = _c(x, np.float32)
    y = np.empty_like(x)
    lib().exact_expf_array(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def expf_np(x):
    x = _c(x, np.float32)
    y = np.empty_like(x)
    lib().exact_expf_np_array(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def logf(x):
    x = _c(x, np.float32)
    y = np.empty_like(x)
    lib().exact_logf_array(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def fix(x, frac):
    x = _c(x, np.float32)
    y = np.empty(x.shape, dtype=np.uint64)
    lib().exact_fix_array(_p(x), frac, _p(y), ctypes.c_int64(x.size))
    return y


def softmax_rows(z, invT):
    z = _c(z, np.float32)
    n, C = z.shape
    p = np.empty_like(z)
    lib().exact_softmax_rows(_p(z), ctypes.c_int64(n), C, ctypes.c_float(invT), _p(p))
    return p


# ------------------------------------------------------------------------------------------------
# stage-1 losses
# ------------------------------------------------------------------------------------------------
LOSS_CE, LOSS_GROUP, LOSS_GROUP_ONLY_MULTI, LOSS_DECOMP, LOSS_TCE = 1, 2, 4, 8, 16


def target_bits(targets, cols_used=None):
    t = _c(targets, np.uint8)
    cols = t.shape[-1]
    cols_used = cols if cols_used is None else cols_used
    bits = np.zeros(t.shape[:-1], dtype=np.uint32)
    lib().exact_target_bits(_p(t), ctypes.c_int64(bits.size), cols, cols_used, _p(bits))
    return bits


def partial_loss_fwd(z, spx, mask, bits, invT, flags):
    """Returns (acc u64[8], gmax u64[N,S,C], losses f32[3])."""
    z = _c(z, np.float32)
    spx = _c(spx, np.int64)
    mask = _c(mask, np.uint8)
    bits = _c(bits, np.uint32)
    N, C, H, W = z.shape
    S = bits.shape[1]
    gmax = np.zeros((N, S, C), dtype=np.uint64)
    acc = np.zeros(8, dtype=np.uint64)
    lib().exact_partial_loss_fwd(_p(z), _p(spx), _p(mask), _p(bits), N, C, H, W, S, ctypes.c_float(invT), flags,
                                 _p(gmax), _p(acc))
    if flags & LOSS_GROUP:
        lib().exact_group_finalize(_p(gmax), ctypes.c_int64(gmax.size), _p(acc))
    losses = np.zeros(3, dtype=np.float32)
    lib().exact_loss_values(_p(acc), flags, _p(losses))
    return acc, gmax, losses


def partial_loss_bwd(z, spx, mask, bits, gmax, acc, grad_out, invT, flags):
    """Returns (scale f32[3], dz f32[N,C,H,W])."""
    z = _c(z, np.float32)
    spx = _c(spx, np.int64)
    mask = _c(mask, np.uint8)
    bits = _c(bits, np.uint32)
    gmax = _c(gmax, np.uint64)
    acc = _c(acc, np.uint64)
    grad_out = _c(grad_out, np.float32)
    N, C, H, W = z.shape
    S = bits.shape[1]
    scale = np.zeros(3, dtype=np.float32)
    lib().exact_loss_scales(_p(acc), _p(grad_out), flags, _p(scale))
    dz = np.empty_like(z)
    lib().exact_partial_loss_bwd(_p(z), _p(spx), _p(mask), _p(bits), _p(gmax), _p(scale), N, C, H, W, S,
                                 ctypes.c_float(invT), flags, _p(dz))
    return scale, dz


def upsample_bilinear(x, Ho, Wo):
    """[..., hi, wi] -> [..., Ho, Wo], F.interpolate(bilinear, align_corners=False) in the kernels' operation order."""
    x = _c(x, np.float32)
    hi, wi = x.shape[-2:]
    nc = int(np.prod(x.shape[:-2]))
    y = np.empty(x.shape[:-2] + (Ho, Wo), dtype=np.float32)
    lib().exact_upsample_bilinear(_p(x), ctypes.c_int64(nc), hi, wi, Ho, Wo, _p(y))
    return y


def partial_loss_bwd_lowres(zq, H, W, spx, mask, bits, gmax, acc, grad_out, invT, flags):
    """Returns (dzq_fix int64 [N,C,h,w], dzq f32): gradient of the losses w.r.t. the quarter-resolution logits."""
    zq = _c(zq, np.float32)
    spx = _c(spx, np.int64)
    mask = _c(mask, np.uint8)
    bits = _c(bits, np.uint32)
    gmax = _c(gmax, np.uint64)
    acc = _c(acc, np.uint64)
    grad_out = _c(grad_out, np.float32)
    N, C, h, w = zq.shape
    S = bits.shape[1]
    scale = np.zeros(3, dtype=np.float32)
    lib().exact_loss_scales(_p(acc), _p(grad_out), flags, _p(scale))
    fix = np.zeros(zq.shape, dtype=np.int64)
    dzq = np.empty_like(zq)
    lib().exact_partial_loss_bwd_lowres(_p(zq), h, w, _p(spx), _p(mask), _p(bits), _p(gmax), _p(scale), N, C, H, W, S,
                                        ctypes.c_float(invT), flags, _p(fix), _p(dzq))
    return fix, dzq


# ------------------------------------------------------------------------------------------------
# single-pass acquisition scan
# ------------------------------------------------------------------------------------------------
def single_pass_accum(z, spx, S, invT):
    z = _c(z, np.float32)
    spx = _c(spx, np.int64)
    B, C, H, W = z.shape
    ps = np.zeros((B, C), dtype=np.uint64)
    cs = np.zeros((B, S, C), dtype=np.uint64)
    hist = np.zeros((B, S, C), dtype=np.uint32)
    lib().exact_single_pass_accum(_p(z), _p(spx), B, C, H, W, S, ctypes.c_float(invT), _p(ps), _p(cs), _p(hist))
    return ps, cs, hist


def weights_to_fixed31(cls_w):
    return np.floor(np.asarray(cls_w, dtype=np.float32).astype(np.float64) * 2147483648.0).astype(np.uint32)


def region_finalize_weighted(class_sum, hist, w31, ban_class):
    class_sum = _c(class_sum, np.uint64)
    hist = _c(hist, np.uint32)
    w31 = _c(w31, np.uint32)
    C = hist.shape[-1]
    shape = hist.shape[:-1]
    n = int(np.prod(shape))
    score = np.zeros(shape, dtype=np.float32)
    dom = np.zeros(shape, dtype=np.int32)
    cnt = np.zeros(shape, dtype=np.uint32)
    lib().exact_region_finalize_weighted(_p(class_sum), _p(hist), ctypes.c_int64(n), C, _p(w31), ban_class,
                                         _p(score), _p(dom), _p(cnt))
    return score, dom, cnt


# ------------------------------------------------------------------------------------------------
# stage-2 cosine pseudo labels
# ------------------------------------------------------------------------------------------------
def stage2_pseudo_labels(feats, logits, targets, spmasks, superpixels, include_onehot, out_hw=None):
    """feats [N,Ch,fh,fw] (full or lower resolution), logits [N,C,H,W] -> int32 [N,H,W] (255 = none)."""
    feats = _c(feats, np.float32)
    logits = _c(logits, np.float32)
    spx = _c(superpixels, np.int64)
    mask = _c(spmasks, np.uint8)
    N, C, H, W = logits.shape
    Ch, fh, fw = feats.shape[1:]
    S = targets.shape[1]
    bits = target_bits(targets)
    flags = LOSS_GROUP | (0 if include_onehot else LOSS_GROUP_ONLY_MULTI)
    out = np.zeros((N, H, W), dtype=np.int32)
    for i in range(N):
        _, gmax, _ = partial_loss_fwd(logits[i:i + 1], spx[i:i + 1], mask[i:i + 1], bits[i:i + 1], np.float32(1.0), flags)
        g = _c(gmax[0], np.uint64)
        lib().exact_stage2_plbl(_p(feats[i]), Ch, fh, fw, H, W, _p(spx[i]), _p(mask[i]), _p(g), S, C, _p(out[i]))
    return out
