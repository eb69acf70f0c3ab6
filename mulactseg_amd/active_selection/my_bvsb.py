"""Best-versus-Second-Best (BvSB) region selector -- reference ``active_selection/my_bvsb.py``.

Per pixel ``p2/p1`` of ``softmax(z/T)``, averaged per superpixel, then min-max normalised over the
pool.  The scan runs in ``k_bvsb_region_accum`` (csrc/scorer.hip); the pool is sharded over ranks.
"""
import torch

from . import base
from .engine import AcquisitionRound
from .utils import get_al_loader


class RegionSelector(base.RegionSelector):
    def __init__(self, args):
        super().__init__(args)
        self.temperature = args.ce_temp

    def _iterate(self, trainer, pool_set, rnd, lowres=False):
        """Yield (first local row, logits, spx) for this rank's reference batches, in loader order.  ``lowres``: the model's
        quarter-resolution logits instead (a consumer that evaluates the final bilinear upsampling itself); the image size is
        then ``spx.shape[-2:]``."""
        model = trainer.net
        model.eval()
        loader, _ = get_al_loader(trainer, pool_set, self.batch_size, self.num_workers, rnd.plan.local_indices)
        row = 0
        with torch.no_grad():
            for batch in loader:
                images = batch['images'].to(trainer.device, dtype=torch.float32)
                spx = batch['spx'].to(trainer.device)
                preds = model(images, lowres=True) if lowres else model(images)
                yield row, preds, spx
                row += images.shape[0]

    def calculate_scores_tensor(self, trainer, pool_set):
        """[n_img, S] f32 on the device: normalised region means (``my_bvsb.py:50-84``)."""
        backend = self._backend(trainer)
        strip = 'predignore' in self.args.method            # my_bvsb.py:65-66: drop the "undefined" channel
        rnd = AcquisitionRound(len(pool_set.im_idx), self.num_class, self.num_superpixels, self.batch_size,
                               self.temperature, backend)
        for row, preds, spx in self._iterate(trainer, pool_set, rnd):
            if strip:
                preds = preds[:, :-1].contiguous()
            if preds.shape[1] != self.num_class:
                raise ValueError("scorer expects %d channels, got %d" % (self.num_class, preds.shape[1]))
            rnd.add_regions(row, preds, spx, None)
        scores = rnd.scores(ban_class=-1)
        backend.minmax_normalize_(scores)                   # my_bvsb.py:79-81
        return scores

    def calculate_scores(self, trainer, pool_set):
        return self.gen_score_list_from_tensor(pool_set, self.calculate_scores_tensor(trainer, pool_set))
