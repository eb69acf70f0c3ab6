"""BvSB + ban of regions dominated by the "undefined" class -- reference
``active_selection/my_bvsb_banignore.py``: all ``num_classes + 1`` channels are scored (no stripping), the
region means are min-max normalised over the pool (:52-56) and then the regions whose arg-max-class histogram
peaks at the last channel are set to 0 (:58-61)."""
from . import my_bvsb
from .engine import AcquisitionRound


class RegionSelector(my_bvsb.RegionSelector):
    extra_channels = 1
    ban_ignore = True
    class_balance = False

    def _scan(self, trainer, pool_set):
        """Unweighted region means + dominant class for every pool image ([n_img,S] f32, [n_img,S] i32)."""
        backend = self._backend(trainer)
        C = self.num_class + self.extra_channels
        if self.extra_channels:
            assert 'predignore' in self.args.method          # my_bvsb_banignore.py:35
        rnd = AcquisitionRound(len(pool_set.im_idx), C, self.num_superpixels, self.batch_size, self.temperature, backend)
        for row, preds, spx in self._iterate(trainer, pool_set, rnd):
            if preds.shape[1] != C:
                raise ValueError("scorer expects %d channels, got %d" % (C, preds.shape[1]))
            rnd.add_regions(row, preds, spx, None)
        self._round = rnd
        return rnd.scores(ban_class=-1, want_dominant=True)

    def _class_weight(self, backend, dominant, C):
        return None

    def calculate_scores_tensor(self, trainer, pool_set):
        backend = self._backend(trainer)
        scores, dominant = self._scan(trainer, pool_set)
        C = self.num_class + self.extra_channels
        backend.minmax_normalize_(scores)                                       # (:52-56)
        w = self._class_weight(backend, dominant, C)
        backend.region_reweight_(scores, dominant.contiguous(), C - 1 if self.ban_ignore else -1, w)
        return scores
