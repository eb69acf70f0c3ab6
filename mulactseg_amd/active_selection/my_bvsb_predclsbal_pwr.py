"""BvSB + pixel-wise class balancing ("PixBal", the proposed acquisition of Hwang et al.) --
reference ``active_selection/my_bvsb_predclsbal_pwr.py`` (VOC form: C = num_classes, no ban).

Default (``single_pass``): ONE scan and ONE model forward per pool image (``k_single_pass``): the class
weight depends only on the pixel's arg-max class, so it factors out of the region sum; the scan keeps per
(region, class) sums of the unweighted margin and the weights are applied in exact integer arithmetic once
the pool's class prior is known.  ``args.two_pass_scoring = True`` selects the reference's own structure:
pass 1 estimates the predicted class prior (K2, ``k_class_prob_sum``), ``k_class_weight`` turns the gathered integer sums into
``cls_weight = (coeff*prior + 1)**-2``; pass 2 averages ``bvsb * cls_weight[top1]`` per superpixel and
histograms the arg-max class (K1+K3, ``k_bvsb_region_accum``).  The two forms agree to ~1e-7 relative (the
per-pixel f32 rounding of ``bvsb * w`` is the only difference) and give the same integers.
"""

from . import my_bvsb
from .. import _lib
from .engine import AcquisitionRound


class RegionSelector(my_bvsb.RegionSelector):
    extra_channels = 0        # C = num_classes (my_bvsb_predclsbal_pwr.py:32,68)
    ban_ignore = False

    def __init__(self, args):
        super().__init__(args)

    def calculate_scores_tensor(self, trainer, pool_set, want_hist=False):
        backend = self._backend(trainer)
        n_img = len(pool_set.im_idx)
        C = self.num_class + self.extra_channels
        ban = C - 1 if self.ban_ignore else -1                                      # (:79-84)
        if not getattr(self.args, 'two_pass_scoring', False):
            rnd = AcquisitionRound(n_img, C, self.num_superpixels, self.batch_size, self.args.ce_temp, backend,
                                   single_pass=True)
            # a model that can hand out its quarter-resolution logits (models/deeplab.py) is scanned from them: the final x4
            # bilinear upsampling happens inside the scan (K8), bit-identical to scanning the upsampled tensor
            low = (getattr(trainer.net, 'lowres_logits', False) and getattr(self.args, 'lowres_scan', True)
                   and C in (19, 20, 21) and hasattr(backend, 'single_pass_lowres'))
            for row, preds, spx in self._iterate(trainer, pool_set, rnd, lowres=low):   # the only pass
                self._check_channels(preds, C)
                if low:
                    try:
                        rnd.add_single_pass_lowres(row, preds, spx.shape[-2:], spx)
                        continue
                    except _lib.MulActSegHipError as e:
                        # the in-kernel upsampling covers ratios >= ~3.8 (the model's is 4); other ratios are refused BEFORE anything
                        # is launched (argument check): scan the upsampled logits instead, for this and the remaining batches
                        if "out of range" not in str(e) and "shape" not in str(e):
                            raise
                        low = False
                        from .. import ops
                        preds = ops.upsample_bilinear(preds.contiguous(), spx.shape[-2:])
                rnd.add_single_pass(row, preds if preds.shape[-2:] == spx.shape[-2:] else self._upsampled(preds, spx), spx)
            cls_w = rnd.class_weights(self.args.cls_weight_coeff)
            self._round, self.cls_weight = rnd, cls_w
            return rnd.scores_single_pass(cls_w, ban_class=ban, want_hist=want_hist)
        rnd = AcquisitionRound(n_img, C, self.num_superpixels, self.batch_size, self.args.ce_temp, backend)
        for row, preds, _ in self._iterate(trainer, pool_set, rnd):                 # pass 1 (:35-43)
            self._check_channels(preds, C)
            rnd.add_prior(row, preds)
        cls_w = rnd.class_weights(self.args.cls_weight_coeff)                       # (:45-47)
        self._round, self.cls_weight = rnd, cls_w
        for row, preds, spx in self._iterate(trainer, pool_set, rnd):               # pass 2 (:49-72)
            rnd.add_regions(row, preds, spx, cls_w)
        return rnd.scores(ban_class=ban, want_hist=want_hist)

    @staticmethod
    def _upsampled(preds, spx):
        from .. import ops
        return ops.upsample_bilinear(preds.contiguous(), spx.shape[-2:])

    @property
    def cumulated_pred_prob(self):
        """Class prior of the last round (``cumulated_pred_prob / len(loader)``, :45), fetched from the device on demand."""
        return self._round.cum

    @staticmethod
    def _check_channels(preds, C):
        if preds.shape[1] != C:
            raise ValueError("model emits %d channels, selector expects %d" % (preds.shape[1], C))
