"""Loss modules of the reference's ``utils/loss.py`` (and the production subclasses defined inside the
reference's trainer files), backed by the HIP kernels of ``csrc/losses.hip``.

Same class names, constructor arguments and ``forward(inputs, targets, superpixels, spmasks)``
signature as the reference, so the trainer plugins read like the reference's:

=============================  =================================================================
class here                     reference definition
=============================  =================================================================
``MyCrossEntropyLoss``         ``utils/loss.py:10-21``
``JointMultiLoss``             ``utils/loss.py:44-63``  ('mean' reduction only, Appendix D)
``GroupMultiLabelCE``          ``utils/loss.py:81-141``  (drops the last target column)
``MultiChoiceCE``              ``utils/loss.py:535-588`` (drops the last target column)
``MultiChoiceCE_``             ``trainer/active_joint_multi_predignore.py:17-73``
``GroupMultiLabelCE_``         ``trainer/active_joint_multi_predignore.py:74-128``
``OnehotCEMultihotChoice``     ``trainer/active_joint_multi_predignore_lossdecomp.py:16-72``
``GroupMultiLabelCE_onlymulti````trainer/active_joint_multi_predignore_mclossablation2.py:17-79``
``FusedPartialLabelLoss``      (new) the three production losses from ONE forward scan and ONE
                               backward scan -- what ``train_impl`` of the production trainer uses
=============================  =================================================================

Inputs must be ROCm tensors; there is no CPU path (``_lib.MulActSegHipError`` otherwise).
"""
import torch
import torch.nn as nn

from .. import _lib, ops


def _fused_forward(ctx, z, size, tgt, cols_used, superpixels, spmasks, invT, flags, sync, weights):
    """Shared forward of the three autograd functions below: ONE library call (prep + scan + finalize with the loss values,
    ops.partial_loss_fwd_fused); ``tgt`` is the u8 multi-hot target tensor (``cols_used`` columns count) or, with ``cols_used``
    None, ready int32 bit masks."""
    z = z.contiguous()
    spx = superpixels.contiguous()
    msk = spmasks.contiguous()
    as_bits = cols_used is None
    losses, st = ops.partial_loss_fwd_fused(z, size, spx, msk, invT, flags, targets=None if as_bits else tgt, cols_used=cols_used,
                                            bits=tgt if as_bits else None, weights=weights, reduce_acc=_all_reduce_sum if sync else None)
    ctx.save_for_backward(z, spx, msk, st.work, tgt if as_bits else st.work, weights if weights is not None else st.work)
    ctx.as_bits, ctx.has_w = as_bits, weights is not None
    ctx.geom = (st.N, st.S, st.C)
    ctx.invT, ctx.flags, ctx.size = invT, flags, size
    ctx.mark_non_differentiable(st.acc)
    return losses, st.acc


def _fused_backward(ctx, grad):
    z, spx, msk, work, bits, weights = ctx.saved_tensors
    st = ops.LossState()
    st.work, st.bits, st.flags = work, bits if ctx.as_bits else None, ctx.flags
    st.N, st.S, st.C = ctx.geom
    return ops.partial_loss_bwd_fused(z, ctx.size, spx, msk, st, grad, ctx.invT, weights=weights if ctx.has_w else None)


class _PartialLossFn(torch.autograd.Function):
    """(ce, mc, group) = f(inputs): the forward direction (accumulators, arg-pixel table, loss values) and the backward direction
    (one scan writing dz) are one library call each.  No host synchronisation in either direction."""

    @staticmethod
    def forward(ctx, inputs, tgt, cols_used, superpixels, spmasks, invT, flags, sync):
        losses, acc = _fused_forward(ctx, inputs, None, tgt, cols_used, superpixels, spmasks, invT, flags, sync, None)
        return losses[0], losses[1], losses[2], acc

    @staticmethod
    def backward(ctx, g_ce, g_mc, g_group, _g_acc):
        grad_out = torch.stack([g_ce, g_mc, g_group]).to(torch.float32).contiguous()
        return _fused_backward(ctx, grad_out), None, None, None, None, None, None, None


class _PartialLossLowResFn(torch.autograd.Function):
    """(ce, mc, group) = f(zq) with the x4 bilinear upsampling of the logits evaluated inside both scans: the
    full-resolution logits and their gradient never exist (``csrc/losses.hip`` LOWRES kernels).  Forward values are
    bit-identical to ``_PartialLossFn`` on ``F.interpolate(zq, size, 'bilinear')``; the gradient of ``zq`` is an
    order-independent fixed-point sum (run-to-run identical)."""

    @staticmethod
    def forward(ctx, zq, size, tgt, cols_used, superpixels, spmasks, invT, flags, sync):
        losses, acc = _fused_forward(ctx, zq, (int(size[0]), int(size[1])), tgt, cols_used, superpixels, spmasks, invT, flags, sync, None)
        return losses[0], losses[1], losses[2], acc

    @staticmethod
    def backward(ctx, g_ce, g_mc, g_group, _g_acc):
        grad_out = torch.stack([g_ce, g_mc, g_group]).to(torch.float32).contiguous()
        return _fused_backward(ctx, grad_out), None, None, None, None, None, None, None, None


class _WeightedLowResLossFn(torch.autograd.Function):
    """total, ce, mc, group = f(zq): the trainer's objective ``w_ce*ce + w_mc*mc + w_group*group`` as ONE differentiable
    output (its value is formed by the last workgroup of the group finalize, its chain rule inside the backward scan), the three
    parts as detached values for logging.  Three launches forward (prep, scan, finalize), three backward (memset, scan, fixed point ->
    float), one library call each."""

    @staticmethod
    def forward(ctx, zq, size, tgt, cols_used, superpixels, spmasks, invT, flags, sync, weights):
        losses, acc = _fused_forward(ctx, zq, (int(size[0]), int(size[1])), tgt, cols_used, superpixels, spmasks, invT, flags, sync, weights)
        parts = losses.detach()
        ctx.mark_non_differentiable(parts)
        return losses[3], parts, acc

    @staticmethod
    def backward(ctx, g_total, _g_parts, _g_acc):
        g = g_total.reshape(1).to(torch.float32).contiguous()
        return _fused_backward(ctx, g), None, None, None, None, None, None, None, None, None


def _all_reduce_sum(acc):
    """Sum the fixed-point loss sums and pixel counts over the data-parallel ranks (RCCL all-reduce of 8
    int64 words): integer addition, so the result does not depend on the number of GPUs."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)


def _run(inputs, targets, superpixels, spmasks, temp, flags, drop_last_column, sync=True):
    """CONTRACT of the drop-in modules under torch.distributed: with ``sync`` (the default of every partial-label module below,
    attribute ``sync_normalisers`` where it is configurable) the returned loss is the objective over the GLOBAL batch, identical
    on every rank; a caller that lets DistributedDataParallel average the gradients must multiply it by the world size before
    ``backward()`` (``ActiveTrainer.update`` does, guarded by ``loss_is_global``).  With ``sync=False`` the loss is the local
    objective and must NOT be scaled.

    ``sync``: under torch.distributed (world > 1) the integer sums and counts are all-reduced before the division, so
    every rank holds the SAME loss value -- the loss over the global batch.  The reference's skip-on-zero / raise-on-NaN
    decisions (``active_joint_multi.py:31-37``) are then global by construction: no rank can skip ``backward()`` while
    its peers wait in the gradient all-reduce."""
    if targets.dtype != torch.uint8:
        targets = targets.to(torch.uint8)
    cols = targets.shape[-1]
    return _PartialLossFn.apply(inputs, targets.contiguous(), cols - 1 if drop_last_column else cols, superpixels, spmasks,
                                ops.inv_temperature(temp), flags, sync)


class MyCrossEntropyLoss(nn.CrossEntropyLoss):
    """Cross entropy with a temperature (stage-2 training) -- reference ``utils/loss.py:10-21``.

    On the GPU (mean reduction, no class weights, no label smoothing, <= 32 channels) the value and the gradient come from the
    partial-label scans in their MAS_LOSS_TCE form: one forward scan and one backward scan over the valid pixels instead of
    ATen's div + log_softmax + nll passes over the materialised ``[N,C,H,W]`` tensor; ``forward_lowres`` takes the model's
    quarter-resolution logits (``net(images, lowres=True)``) and evaluates the x4 bilinear upsampling
    (``models/segmentation/utils.py:25``) inside the scans -- the full-resolution logits and their gradient never exist.
    Under torch.distributed the sums and the valid-pixel counts are all-reduced before the division when ``sync_normalisers``
    (the mean over the GLOBAL batch; multiply by the world size before ``backward()`` under DistributedDataParallel)."""

    def __init__(self, ignore_index, reduction='mean', temperature=1.0, sync_normalisers=False):
        super().__init__(ignore_index=ignore_index, reduction=reduction)
        self.temperature = temperature
        self.sync_normalisers = sync_normalisers

    def _fused(self, input, target):
        return (input.is_cuda and self.reduction == 'mean' and self.weight is None and getattr(self, 'label_smoothing', 0.0) == 0.0
                and input.dim() == 4 and input.shape[1] <= _lib.MAX_CLASSES and input.dtype == torch.float32
                and target.dim() == 3 and not target.is_floating_point())

    def _labels_mask(self, input, target):
        C = input.shape[1]
        labels = target.contiguous()
        # (labels outside [0, C) other than ignore_index are an error in torch; here they are left out like ignored ones)
        mask = (labels != self.ignore_index) & (labels >= 0) & (labels < C)
        dummy_bits = torch.zeros((input.shape[0], C), dtype=torch.int32, device=input.device)
        return labels, mask, dummy_bits

    def forward(self, input, target):
        if not self._fused(input, target):
            return super().forward(input / self.temperature, target)
        labels, mask, bits = self._labels_mask(input, target)
        ce, _, _, self.last_acc = _PartialLossFn.apply(input, bits, None, labels, mask, ops.inv_temperature(self.temperature),
                                                        _lib.LOSS_CE | _lib.LOSS_TCE, self.sync_normalisers)
        return ce

    def forward_lowres(self, quarter_logits, size, target):
        """CE of ``F.interpolate(quarter_logits, size, 'bilinear', align_corners=False) / T`` against ``target`` [N,H,W]."""
        if not self._fused(quarter_logits, target):
            up = torch.nn.functional.interpolate(quarter_logits, size=tuple(size), mode='bilinear', align_corners=False)
            return super().forward(up / self.temperature, target)
        labels, mask, bits = self._labels_mask(quarter_logits, target)
        ce, _, _, self.last_acc = _PartialLossLowResFn.apply(quarter_logits, tuple(size), bits, None, labels, mask,
                                                             ops.inv_temperature(self.temperature), _lib.LOSS_CE | _lib.LOSS_TCE,
                                                             self.sync_normalisers)
        return ce


class MultiChoiceCE(nn.Module):
    """Merged-positive CE -- reference ``utils/loss.py:535-588`` (last target column dropped)."""
    _drop_last = True
    _flags = _lib.LOSS_CE

    def __init__(self, num_class, temperature=1.0, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' is on the hot path")
        self.num_class = num_class
        self.reduction = reduction
        self.eps = 1e-8
        self.temp = temperature

    def forward(self, inputs, targets, superpixels, spmasks):
        ce, _, _, _ = _run(inputs, targets, superpixels, spmasks, self.temp, self._flags, self._drop_last)
        return ce


class MultiChoiceCE_(MultiChoiceCE):
    """All target columns (incl. "undefined") -- ``trainer/active_joint_multi_predignore.py:17-73``."""
    _drop_last = False


class OnehotCEMultihotChoice(MultiChoiceCE):
    """Returns (ce over one-hot superpixels, mc over multi-hot superpixels) --
    ``trainer/active_joint_multi_predignore_lossdecomp.py:16-72``."""
    _drop_last = False
    _flags = _lib.LOSS_CE | _lib.LOSS_DECOMP

    def forward(self, inputs, targets, superpixels, spmasks):
        ce, mc, _, _ = _run(inputs, targets, superpixels, spmasks, self.temp, self._flags, self._drop_last)
        return ce, mc


class GroupMultiLabelCE(nn.Module):
    """Group / MIL loss: -log of the per-superpixel class-wise max probability --
    reference ``utils/loss.py:81-141`` (last target column dropped)."""
    _drop_last = True
    _flags = _lib.LOSS_GROUP

    def __init__(self, args, num_class, num_superpixel, temperature=1.0, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' is on the hot path")
        self.args = args
        self.num_class = num_class
        self.num_superpixel = num_superpixel
        self.eps = 1e-8
        self.temp = temperature
        self.reduction = reduction

    def forward(self, inputs, targets, superpixels, spmasks):
        if targets.shape[1] != self.num_superpixel:
            raise ValueError("targets carry %d superpixels, loss was built for %d" % (targets.shape[1], self.num_superpixel))
        _, _, group, _ = _run(inputs, targets, superpixels, spmasks, self.temp, self._flags, self._drop_last)
        return group


class GroupMultiLabelCE_(GroupMultiLabelCE):
    """``trainer/active_joint_multi_predignore.py:74-128``."""
    _drop_last = False


class GroupMultiLabelCE_onlymulti(GroupMultiLabelCE_):
    """Only superpixels with more than one target bit --
    ``trainer/active_joint_multi_predignore_mclossablation2.py:17-79``."""
    _flags = _lib.LOSS_GROUP | _lib.LOSS_GROUP_ONLY_MULTI


class JointMultiLoss(nn.Module):
    """``utils/loss.py:44-63`` ('mean' path): returns (loss_group, loss_pos)."""

    def __init__(self, group_multi_loss, multi_pos_loss, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("the reference's 'none' branch is broken (wrong arity, utils/loss.py:57-61)")
        self.group_multi_loss = group_multi_loss
        self.multi_pos_loss = multi_pos_loss
        self.reduction = reduction

    def forward(self, inputs, targets, superpixels, spmasks):
        return (self.group_multi_loss(inputs, targets, superpixels, spmasks),
                self.multi_pos_loss(inputs, targets, superpixels, spmasks))


class FusedPartialLabelLoss(nn.Module):
    """The production stage-1 objective from one forward scan and one backward scan:
    returns (group_loss, ce_loss, mc_loss) exactly as
    ``GroupMultiLabelCE_onlymulti`` + ``OnehotCEMultihotChoice`` would
    (``trainer/active_joint_multi_predignore_lossdecomp.py:102-103``), reading the logits once."""

    def __init__(self, num_superpixel, group_temperature=1.0, multi_temperature=1.0, only_multi=True, decomp=True,
                 sync_normalisers=True):
        super().__init__()
        # under torch.distributed: all-reduce (sum, n) before the division so that the objective equals the
        # single-GPU objective over the global batch; multiply the loss by world_size before backward()
        # when gradients are then AVERAGED over ranks (DistributedDataParallel does).
        self.sync_normalisers = sync_normalisers
        if group_temperature != multi_temperature:
            raise ValueError("the fused scan shares one softmax: group_ce_temp must equal multi_ce_temp "
                             "(both 0.1 in the reference scripts)")
        self.num_superpixel = num_superpixel
        self.temp = multi_temperature
        self.flags = _lib.LOSS_CE | _lib.LOSS_GROUP | (_lib.LOSS_GROUP_ONLY_MULTI if only_multi else 0) \
            | (_lib.LOSS_DECOMP if decomp else 0)

    def forward(self, inputs, targets, superpixels, spmasks):
        ce, mc, group, self.last_acc = _run(inputs, targets, superpixels, spmasks, self.temp, self.flags, False,
                                            sync=self.sync_normalisers)
        return group, ce, mc

    def forward_lowres(self, quarter_logits, size, targets, superpixels, spmasks):
        """The same three losses from the model's quarter-resolution logits (``net(images, lowres=True)``): the final x4
        bilinear upsampling (``models/segmentation/utils.py:25``) happens per selected pixel inside the scans."""
        if targets.dtype != torch.uint8:
            targets = targets.to(torch.uint8)
        ce, mc, group, self.last_acc = _PartialLossLowResFn.apply(quarter_logits, tuple(size), targets.contiguous(), targets.shape[-1], superpixels, spmasks,
                                                                  ops.inv_temperature(self.temp), self.flags, self.sync_normalisers)
        return group, ce, mc

    def weighted_lowres(self, quarter_logits, size, targets, superpixels, spmasks, coeff, coeff_mc, coeff_gm):
        """``(coeff * ce + coeff_mc * mc) + coeff_gm * group`` (``..._lossdecomp.py:104``) as one differentiable scalar, plus the
        detached parts ``(group, ce, mc)`` for logging -- same value, bit for bit, as composing ``forward_lowres`` with torch
        arithmetic, in half the kernel launches."""
        if targets.dtype != torch.uint8:
            targets = targets.to(torch.uint8)
        key = (float(coeff), float(coeff_mc), float(coeff_gm), quarter_logits.device)
        if getattr(self, '_w_key', None) != key:
            self._w_key, self._w = key, torch.tensor(key[:3], dtype=torch.float32, device=quarter_logits.device)
        total, parts, self.last_acc = _WeightedLowResLossFn.apply(quarter_logits, tuple(size), targets.contiguous(), targets.shape[-1], superpixels, spmasks,
                                                                  ops.inv_temperature(self.temp), self.flags, self.sync_normalisers, self._w)
        return total, parts[2], parts[0], parts[1]
