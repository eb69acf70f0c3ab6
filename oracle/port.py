"""CPU restatement ("port") of the MulActSeg hot path in plain PyTorch-CPU.

ORACLE / TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under ``mulactseg_amd/`` does.

What it is: the reference's algorithm for the path, written from scratch against the reference's
behaviour, with the same f32 operation order wherever that order is visible in the results (so that
it reproduces the golden vectors that ``oracle/gen_golden.py`` produced by *running the reference's
own Python* in the build container).  It does not depend on ``torch_scatter``: the three segment
reductions the reference takes from that package are restated below (published semantics of
pytorch-scatter 2.0.9, pinned in ``actsegmul.yml:99``).

Pinning: ``tests/test_oracle_golden.py`` checks every function here against ``tests/golden/*.npz``.
The reference repository itself holds no tests or golden vectors for this path,
so the goldens generated from its executed code are the only pin.

All ``file:line`` citations are relative to the reference repository root.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-8


# ----------------------------------------------------------------------------------------------
# segment reductions (third-party torch_scatter 2.0.9 semantics, restated)
# ----------------------------------------------------------------------------------------------
def segment_sum(src, index, dim_size):
    """``scatter(src, index, dim=-2|-1, reduce='sum')`` for src [..., n] / [..., n, C], index [..., n].

    CPU semantics: ``zeros.scatter_add_`` -- sequential accumulation in source order.
    """
    if src.dim() == index.dim():
        out = torch.zeros(src.shape[:-1] + (dim_size,), dtype=src.dtype)
        return out.scatter_add_(-1, index, src)
    idx = index.unsqueeze(-1).expand(src.shape)
    out = torch.zeros(src.shape[:-2] + (dim_size, src.shape[-1]), dtype=src.dtype)
    return out.scatter_add_(-2, idx, src)


def segment_mean(src, index, dim_size):
    """``scatter(..., reduce='mean')``: sum / count, count accumulated in src.dtype and clamped >= 1."""
    out = segment_sum(src, index, dim_size)
    count = segment_sum(torch.ones(index.shape, dtype=src.dtype), index, dim_size)
    count[count < 1] = 1
    return out.true_divide_(count)


class _SegmentMax(torch.autograd.Function):
    """``scatter(src[n,C], index[n], dim=0, reduce='max', dim_size=S)`` -> (out[S,C], arg[S,C]).

    Rows that receive nothing: out = 0, arg = n.  Ties: first source row wins.  Backward: gradient
    goes to the arg row only.
    """

    @staticmethod
    def forward(ctx, src, index, dim_size):
        n, C = src.shape
        idx = index.view(-1, 1).expand(n, C)
        out = torch.full((dim_size, C), torch.finfo(src.dtype).min, dtype=src.dtype)
        out.scatter_reduce_(0, idx, src, 'amax', include_self=True)
        pos = torch.arange(n).view(-1, 1).expand(n, C)
        cand = torch.where(src == out.gather(0, idx), pos, torch.full_like(pos, n))
        arg = torch.full((dim_size, C), n, dtype=torch.long)
        arg.scatter_reduce_(0, idx, cand, 'amin', include_self=True)
        out = out.masked_fill(arg == n, 0)
        ctx.n = n
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _):
        (arg,) = ctx.saved_tensors
        g = torch.zeros((ctx.n + 1, grad_out.shape[1]), dtype=grad_out.dtype)
        g.scatter_(0, arg, grad_out)
        return g[:ctx.n], None, None


def segment_max(src, index, dim_size):
    return _SegmentMax.apply(src, index, dim_size)


# ----------------------------------------------------------------------------------------------
# acquisition scorer (K1-K4)
# ----------------------------------------------------------------------------------------------
def softmax_bvsb(preds, temperature):
    """Per-pixel Best-vs-Second-Best margin and arg-max class.

    Follows ``active_selection/my_bvsb.py:19-27``: softmax(preds / T) over dim 1, top-2, ratio
    second/first, ``+= 1e-8``; top1 = index of the best.  (On exact top-2 ties ``torch.topk`` does not
    define which index is first; the HIP path and ``oracle/exact`` define lowest-index-wins.)
    """
    prob = torch.softmax(preds / temperature, dim=1)
    val, idx = torch.topk(prob, 2, dim=1)
    bvsb = val[:, 1] / val[:, 0]
    bvsb += EPS
    return bvsb, idx[:, 0]


def class_prior_batch(preds, ce_temp):
    """One batch's contribution to the predicted class prior:
    ``torch.mean(softmax(preds / ce_temp, dim=1), dim=(0, 2, 3))`` --
    ``active_selection/my_bvsb_predclsbal_pwr_banignore.py:41-42``."""
    return torch.mean(torch.softmax(preds / ce_temp, dim=1), dim=(0, 2, 3))


def class_weight(batch_means, cls_weight_coeff):
    """``cum = sum_b mean_b / n_batches`` (mean of per-batch means -- a short last batch is
    over-weighted, deliberately replicated) then ``(coeff * cum + 1) ** -2`` --
    ``my_bvsb_predclsbal_pwr_banignore.py:33,42,45,47``."""
    cum = torch.zeros_like(batch_means[0])
    for m in batch_means:
        cum += m
    cum = cum / len(batch_means)
    return cum, (cls_weight_coeff * cum + 1) ** (-2)


def region_scores_batch(preds, spx, temperature, cls_weight, num_superpixels, n_hist_classes):
    """Pass-2 body for one batch: per-superpixel mean of the class-weighted BvSB and the
    per-superpixel histogram of the arg-max class.

    Follows ``my_bvsb_predclsbal_pwr_banignore.py:57-69`` (VOC twin
    ``my_bvsb_predclsbal_pwr.py:57-69``).  ``cls_weight=None`` gives the unweighted
    ``my_bvsb.py:66-73`` variant.  Returns (region_bvsb [B,S] f32, region_ntop1 [B,S,C] i64).
    """
    bvsb, top1 = softmax_bvsb(preds, temperature)
    B, H, W = top1.shape
    if cls_weight is not None:
        bvsb = bvsb * cls_weight[top1.reshape(-1)].view(B, H, W)
    ids = spx.view(B, -1)
    region_bvsb = segment_mean(bvsb.view(B, -1), ids, num_superpixels)
    # scatter-sum of one_hot(top1): identical integers to the reference's materialised one-hot
    flat = ids * n_hist_classes + top1.view(B, -1)
    hist = torch.zeros((B, num_superpixels * n_hist_classes), dtype=torch.long)
    hist.scatter_add_(1, flat, torch.ones_like(flat))
    return region_bvsb, hist.view(B, num_superpixels, n_hist_classes)


def ban_ignore_dominant(scores_flat, hist_flat):
    """``dominant = argmax(hist)``; regions dominated by the last ("undefined") channel get score 0 --
    ``my_bvsb_predclsbal_pwr_banignore.py:79-84``."""
    dominant = hist_flat.argmax(dim=1)
    scores_flat = scores_flat.clone()
    scores_flat[dominant == hist_flat.shape[1] - 1] = 0
    return scores_flat, dominant


def _batches(n, bs):
    return [(i, min(i + bs, n)) for i in range(0, n, bs)]


def pixbal_scores(logits, spx, batch_size, ce_temp, cls_weight_coeff, num_superpixels, ban_ignore):
    """Whole ``calculate_scores`` of the proposed PixBal selector on resident logits.

    ``logits`` [N,C,H,W] f32 stands for ``model(images)``; images are visited in loader order in
    batches of ``batch_size`` (``active_selection/utils.py:47-57``, shuffle=False).
    ``ban_ignore=True``  -> ``my_bvsb_predclsbal_pwr_banignore.py:22-91`` (C = num_classes + 1),
    ``ban_ignore=False`` -> ``my_bvsb_predclsbal_pwr.py:23-88``.
    Returns dict(cum, cls_weight, region_bvsb [N,S], region_ntop1 [N,S,C], scores [N,S]).
    """
    N, C = logits.shape[:2]
    means = [class_prior_batch(logits[a:b], ce_temp) for a, b in _batches(N, batch_size)]
    cum, w = class_weight(means, cls_weight_coeff)
    rb, rh = [], []
    for a, b in _batches(N, batch_size):
        r, h = region_scores_batch(logits[a:b], spx[a:b], ce_temp, w, num_superpixels, C)
        rb.append(r)
        rh.append(h)
    rb, rh = torch.cat(rb), torch.cat(rh)
    scores = rb.view(-1)
    if ban_ignore:
        scores, _ = ban_ignore_dominant(scores, rh.view(-1, C))
    return dict(cum=cum, cls_weight=w, region_bvsb=rb, region_ntop1=rh,
                scores=scores.view(-1, num_superpixels))


def bvsb_scores(logits, spx, batch_size, temperature, num_superpixels, strip_last):
    """``my_bvsb.py:50-87``: unweighted region means, then min-max normalisation over the pool:
    ``u -= min(u[u != 0]); u /= max(u)`` (absent regions become negative)."""
    if strip_last:                       # 'predignore' in args.method, my_bvsb.py:65-66
        logits = logits[:, :-1]
    N = logits.shape[0]
    rb = []
    for a, b in _batches(N, batch_size):
        r, _ = region_scores_batch(logits[a:b], spx[a:b], temperature, None, num_superpixels,
                                   logits.shape[1])
        rb.append(r)
    u = torch.cat(rb).view(-1)
    u = u - u[u != 0].min()
    u = u / u.max()
    return u.view(-1, num_superpixels)


def bvsb_variant_scores(logits, spx, batch_size, temperature, num_superpixels, ban_ignore, class_balance):
    """The remaining BvSB selectors: unweighted region means over ALL channels, min-max normalisation, then
    optionally the ban of "undefined"-dominated regions (``my_bvsb_banignore.py:52-61``) and the region-level
    class balancing ``exp(-share of regions with that dominant class)`` (``my_bvsb_clsbal_v2[_banignore].py:60-74``)."""
    N, C = logits.shape[:2]
    rb, rh = [], []
    for a, b in _batches(N, batch_size):
        r, h = region_scores_batch(logits[a:b], spx[a:b], temperature, None, num_superpixels, C)
        rb.append(r)
        rh.append(h)
    u = torch.cat(rb).view(-1)
    hist = torch.cat(rh).view(-1, C)
    u = u - u[u != 0].min()
    u = u / u.max()
    dominant = hist.argmax(dim=1)
    if ban_ignore:
        u[dominant == C - 1] = 0
    w = None
    if class_balance:
        oh = F.one_hot(dominant, num_classes=C)
        dist = oh.sum(dim=0) / oh.sum()
        w = torch.exp(-dist)
        u = w[dominant] * u
    return u.view(-1, num_superpixels), w


def score_list(im_idx, suppix, scores):
    """``gen_score_list_from_tensor`` -- ``my_bvsb.py:29-48``: (score, "img,lbl,spx", id) for every id
    still listed in ``suppix[spx_path]``, images in ``im_idx`` order."""
    out = []
    for k, key in enumerate(im_idx):
        path = ','.join(key)
        ids = suppix[key[2]]
        vals = scores[k][ids].tolist()
        out.extend((s, path, i) for s, i in zip(vals, ids))
    return out


def select_regions(score_tuples, budget, cost_fn=None):
    """``sorted(scores, reverse=True)`` (``active_selection/base.py:37``) then the budget walk of
    ``dataloader/region_active_dataset.py:31-73``: take regions in order, cost += cost_fn(path, id)
    (1 when ``cost_fn`` is None), stop after the region that makes ``cost > budget``.
    Returns the consumed prefix of the sorted list."""
    ordered = sorted(score_tuples, reverse=True)
    cost = 0
    for n, (_, path, rid) in enumerate(ordered):
        cost += 1 if cost_fn is None else cost_fn(path, rid)
        if cost > budget:
            return ordered[:n + 1]
    return ordered


# ----------------------------------------------------------------------------------------------
# stage-1 partial-label losses (K5 / K6)
# ----------------------------------------------------------------------------------------------
def merged_positive_ce(inputs, targets, superpixels, spmasks, temp, variant):
    """Merged-positive ("multi-choice") CE over the selected pixels.

    variant 'decomp'     -> ``OnehotCEMultihotChoice.forward``
                            (``trainer/active_joint_multi_predignore_lossdecomp.py:21-72``):
                            returns (ce over one-hot regions, mc over multi-hot regions).
    variant 'predignore' -> ``MultiChoiceCE_.forward`` (``active_joint_multi_predignore.py:21-73``):
                            one sum over pixels with >= 1 target bit, all C columns.
    variant 'base'       -> ``MultiChoiceCE.forward`` (``utils/loss.py:543-588``): as above but the
                            last target column is dropped.
    Per pixel: pos = sum_c softmax(z/T)_c * Y_c ; l = -log(pos + 1e-8); normaliser 1 + n.
    """
    N, C, H, W = inputs.shape
    x = inputs.permute(0, 2, 3, 1).reshape(N, -1, C)
    out = F.softmax(x / temp, dim=2)
    spx = superpixels.reshape(N, -1)
    msk = spmasks.reshape(N, -1)
    sums = [0, 0]
    cnts = [1, 1]
    for i in range(N):
        m = msk[i]
        if not torch.any(m):
            continue
        vo = out[i][m]
        trg = targets[i] if variant != 'base' else targets[i][..., :-1]
        tp = trg[spx[i][m]]
        if variant == 'decomp':
            pos = (vo * tp).sum(dim=1)
            nb = tp.sum(dim=1)
            one = nb == 1
            if torch.any(one):
                v = pos[one]
                sums[0] = sums[0] + (-torch.log(v + EPS).sum())
                cnts[0] += v.shape[0]
            mul = torch.logical_not(one)
            if torch.any(mul):
                assert torch.all(mul == (1 < nb)), "selected superpixel without any target bit"
                v = pos[mul]
                sums[1] = sums[1] + (-torch.log(v + EPS).sum())
                cnts[1] += v.shape[0]
        else:
            keep = torch.any(tp, dim=1).bool()
            pos = (vo[keep] * tp[keep]).sum(dim=1)
            cnts[0] += pos.shape[0]
            sums[0] = sums[0] + (-torch.log(pos + EPS).sum())
    if variant == 'decomp':
        return sums[0] / cnts[0], sums[1] / cnts[1]
    return sums[0] / cnts[0]


def group_max_ce(inputs, targets, superpixels, spmasks, num_superpixel, temp, variant):
    """Group / MIL loss: per selected superpixel s and class c in Y_s:
    ``-log(max_{p in s} softmax(z_p/T)_c + 1e-8)``, normaliser 1 + #nonzero.

    variant 'onlymulti'  -> ``GroupMultiLabelCE_onlymulti.forward``
                            (``trainer/active_joint_multi_predignore_mclossablation2.py:22-79``):
                            only superpixels with > 1 target bit.
    variant 'predignore' -> ``GroupMultiLabelCE_.forward`` (``active_joint_multi_predignore.py:82-128``).
    variant 'base'       -> ``GroupMultiLabelCE.forward`` (``utils/loss.py:91-141``): last target
                            column dropped.
    """
    N, C, H, W = inputs.shape
    out = F.softmax(inputs / temp, dim=1).permute(0, 2, 3, 1).reshape(N, -1, C)
    spx = superpixels.reshape(N, -1)
    msk = spmasks.reshape(N, -1)
    tg = targets if variant != 'base' else targets[..., :-1]
    nonempty = torch.any(tg, dim=2).bool()
    is_multi = 1 < targets.sum(dim=2)
    loss = 0
    num_valid = 1
    for i in range(N):
        m = msk[i]
        if not torch.any(m):
            continue
        if variant == 'onlymulti':
            m = m & is_multi[i][spx[i].clamp(max=num_superpixel - 1)]
            # (pad pixels carry id == num_superpixel but are never selected, so the clamp only keeps
            #  the gather in range for masked-out pixels; the reference gathers selected pixels only)
            if not torch.any(m):
                continue
        pooled, _ = segment_max(out[i][m], spx[i][m], num_superpixel)
        pooled = pooled[nonempty[i]]
        trg = tg[i][nonempty[i]]
        top = pooled * trg
        nz = top[top.nonzero(as_tuple=True)]
        num_valid += nz.shape[0]
        loss = loss + (-torch.log(nz + EPS).sum())
    return loss / num_valid


def temperature_ce(inputs, target, temperature, ignore_index=255):
    """``MyCrossEntropyLoss`` -- ``utils/loss.py:10-21``: CE(z / T, y), mean over non-ignored."""
    return F.cross_entropy(inputs / temperature, target, ignore_index=ignore_index)


# ----------------------------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------------------------
def iou_counts(outputs, targets, num_classes, ignore_label):
    """seen / correct / positive per class over non-ignored pixels -- ``utils/miou.py:23-38``."""
    keep = targets != ignore_label
    o, t = outputs[keep], targets[keep]
    seen = np.zeros(num_classes)
    correct = np.zeros(num_classes)
    positive = np.zeros(num_classes)
    for i in range(num_classes):
        seen[i] = torch.sum(t == i).item()
        correct[i] = torch.sum((t == i) & (o == t)).item()
        positive[i] = torch.sum(o == i).item()
    return seen, correct, positive


def ious_from_counts(seen, correct, positive):
    """``MeanIoU._after_epoch`` -- ``utils/miou.py:57-71``: unseen class counts as 1 (-> 100)."""
    ious = []
    for s, c, p in zip(seen, correct, positive):
        ious.append(1 if s == 0 else c / (s + p - c))
    return [v * 100 for v in ious]


def ignore_iou_counts(outputs, targets, num_classes, ignore_label):
    """``IoUIgnore._after_step`` -- ``utils/miou_evalignore.py:20-32``."""
    seen = torch.sum(targets == ignore_label).item()
    correct = torch.sum((targets == ignore_label) & (outputs == num_classes)).item()
    positive = torch.sum(outputs == num_classes).item()
    return seen, correct, positive


# ----------------------------------------------------------------------------------------------
# stage-2: cosine pseudo-label generation with one-ring propagation (K9)
# ----------------------------------------------------------------------------------------------
def cosine_pseudo_labels(feats, logits, targets, spmasks, superpixels, nseg, include_onehot, threshold_method='median'):
    """Pseudo labels from class prototypes inside and one superpixel ring around the labelled regions.

    Follows ``trainer/eval_save_cosplbl_prop.py:121-314`` (``include_onehot=False``: only superpixels with more than
    one target bit) and ``..._includeonehot.py`` (every selected pixel).  Per image:
      1. p = softmax(logits) (no temperature, :140); valid pixels = selected (and multi-hot) pixels;
      2. for every valid superpixel s and class c in Y_s the PROTOTYPE is the feature of the valid pixel of s with
         the largest p_c (``scatter_max``, first index on ties) (:173-199);
      3. every valid pixel takes the class of the most similar prototype of ITS OWN superpixel (:203-232, :310-311);
      4. per prototype the similarity threshold is the (lower) median of the similarities of the pixels assigned to
         it, 1.0 if none (:234-257);
      5. each valid superpixel, in ascending id order, labels the pixels of its 3x3-dilation neighbours (itself
         included) with the class of their most similar prototype wherever some prototype's similarity exceeds that
         prototype's threshold; later superpixels overwrite earlier ones (:259-306); step 3 is written last.
    Returns int64 [N,H,W], 255 = no pseudo label."""
    from scipy import ndimage
    N, C, H, W = logits.shape
    Ch = feats.shape[1]
    out = torch.full((N, H * W), 255, dtype=torch.long)
    is_multi = 1 < targets.sum(dim=2)
    for i in range(N):
        p = F.softmax(logits[i:i + 1], dim=1)[0].permute(1, 2, 0).reshape(-1, C)
        fe = feats[i].permute(1, 2, 0).reshape(-1, Ch)
        sp = superpixels[i].reshape(-1)
        valid = spmasks[i].reshape(-1).clone()
        if not torch.any(valid):
            continue
        if not include_onehot:
            valid = valid & is_multi[i][sp.clamp(max=nseg - 1)]
            if not torch.any(valid):
                continue
        vpix = valid.nonzero().squeeze(1)
        vsp = sp[vpix]
        vp, vf = p[vpix], fe[vpix]
        _, arg = segment_max(vp, vsp, nseg)                       # [nseg, C] indices into the valid list
        sp_valid = arg[:, 0] < vp.shape[0]
        sp_ids = sp_valid.nonzero().squeeze(1)                     # ascending superpixel ids
        tg = targets[i][sp_valid]
        proto_g, proto_c = tg.nonzero(as_tuple=True)               # ordered by (superpixel, class)
        proto_v = arg[sp_valid][proto_g, proto_c]
        protos = vf[proto_v]
        sim = torch.mm(protos, vf.T)                               # [n_proto, n_valid]
        best_sim, best_proto = segment_max(sim, proto_g, int(proto_g.max()) + 1)   # per group, per valid pixel
        to_group = torch.full((nseg,), -1, dtype=torch.long)
        to_group[sp_ids] = torch.arange(sp_ids.shape[0])
        vgroup = to_group[vsp]
        cols = torch.arange(vgroup.shape[0])
        nn_proto = best_proto.T[cols, vgroup]
        nn_sim = best_sim.T[cols, vgroup]
        # thresholds per prototype
        thr = torch.ones(protos.shape[0])
        for k in range(protos.shape[0]):
            sel = nn_sim[nn_proto == k]
            if sel.numel():
                thr[k] = torch.median(sel) if threshold_method == 'median' else torch.min(sel)
        # one-ring propagation, ascending superpixel id, later overwrites earlier
        spmap = superpixels[i].numpy()
        for g, s in enumerate(sp_ids.tolist()):
            ring = np.unique(spmap[ndimage.binary_dilation(spmap == s, structure=np.ones((3, 3), np.uint8))])
            around = torch.from_numpy(np.isin(spmap, ring).reshape(-1))
            mine = proto_g == g
            s_around = torch.mm(protos[mine], fe[around].T)        # [n_mine, n_around]
            label = proto_c[mine][s_around.argmax(dim=0)]
            ok = torch.any(thr[mine][:, None] < s_around, dim=0)
            pix = around.nonzero().squeeze(1)
            out[i, pix[ok]] = label[ok]
        out[i, vpix] = proto_c[nn_proto]
    return out.reshape(N, H, W)
