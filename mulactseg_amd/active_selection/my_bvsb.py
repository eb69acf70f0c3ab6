"""Best-versus-Second-Best (BvSB) region selector -- reference ``active_selection/my_bvsb.py``.

Per pixel ``p2/p1`` of ``softmax(z/T)``, averaged per superpixel, then min-max normalised over the
pool.  The scan runs in ``k_bvsb_region_accum`` (csrc/scorer.hip); the pool is sharded over ranks.
"""
import contextlib
import os

import torch

from . import base
from .engine import AcquisitionRound
from .utils import get_al_loader


_POOL_STREAMS = {}


def pool_streams(dev):
    """The streams consecutive pool batches alternate between (MAS_POOL_STREAMS, default 2; 1 = the caller's stream only); [] on the
    CPU (tests with the oracle-backed stand-in)."""
    n = int(os.environ.get("MAS_POOL_STREAMS", "2"))
    if dev.type != 'cuda' or n <= 1:
        return []
    key = (dev, n)
    if key not in _POOL_STREAMS:
        _POOL_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _POOL_STREAMS[key]


class RegionSelector(base.RegionSelector):
    def __init__(self, args):
        super().__init__(args)
        self.temperature = args.ce_temp

    def _iterate(self, trainer, pool_set, rnd, lowres=False):
        """Yield (first local row, logits, spx) for this rank's reference batches, in loader order.  ``lowres``: the model's
        quarter-resolution logits instead (a consumer that evaluates the final bilinear upsampling itself); the image size is
        then ``spx.shape[-2:]``.

        On the GPU consecutive batches alternate between TWO HIP streams (``pool_streams``): the batch is produced, run through the
        model and consumed (the caller's scan runs while this generator is suspended inside the stream context) on one stream while
        the previous batch drains on the other -- the last, partly filled wave of workgroups of every layer of one batch is covered
        by the other batch's kernels: 16.7 -> 16.1 ms per Cityscapes pool batch (profiles/r06).  Every accumulator of the scans is
        an integer sum or a per-picture row (csrc/detmath.h), so the scores do not depend on the interleaving.  Both streams are
        joined into the caller's stream before the generator ends."""
        model = trainer.net
        model.eval()
        loader, _ = get_al_loader(trainer, pool_set, self.batch_size, self.num_workers, rnd.plan.local_indices)
        dev = torch.device(trainer.device)
        streams = pool_streams(dev)
        main = torch.cuda.current_stream(dev) if streams else None
        row, k = 0, 0
        it = iter(loader)
        try:
            with torch.no_grad():
                while True:
                    # The FIRST batch runs on the caller's stream: everything the model derives lazily from its weights (folded
                    # BatchNorm constants, split / packed weight images: ops._bn_fold, ops._conv_bx_weight, ...) is built there, once,
                    # by kernels of that stream; the pool streams start behind it.  (Built lazily on one pool stream, such a constant
                    # would be read by the other stream's kernels before the kernel that writes it has run.)
                    if k == 1:
                        for st in streams:
                            st.wait_stream(main)
                    ctx = torch.cuda.stream(streams[k % len(streams)]) if (streams and k > 0) else contextlib.nullcontext()
                    with ctx:
                        try:
                            batch = next(it)        # (a file-backed / resident pool makes the sample on the current stream: inside the context)
                        except StopIteration:
                            break
                        images = batch['images'].to(trainer.device, dtype=torch.float32)
                        spx = batch['spx'].to(trainer.device)
                        preds = model(images, lowres=True) if lowres else model(images)
                        yield row, preds, spx
                        row += images.shape[0]
                        del batch, images, spx, preds       # (freed into the pool of the stream that used them)
                    k += 1
        finally:
            for st in streams:
                main.wait_stream(st)

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
