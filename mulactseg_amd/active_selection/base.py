"""Base region selector -- reference ``active_selection/base.py:13-38``.

``select_next_batch`` keeps the reference's contract (``calculate_scores`` -> sort descending ->
``active_set.expand_training_set``) but, for selectors that can score on the device
(``calculate_scores_tensor``), does the ordering and the budget walk on the GPU (K4) and hands
``expand_training_set`` only the consumed prefix -- the same prefix the reference would consume, so the
resulting active set, ``isselected`` matrix and selection pickle are identical, without ever
materialising (or Python-sorting) the ~6 M tuples of the Cityscapes pool.
"""
import json
import os

import numpy as np
import torch


class RegionSelector(object):

    def __init__(self, args):
        self.args = args
        self.batch_size = args.val_batch_size
        self.num_workers = args.val_num_workers
        self.num_superpixels = args.nseg
        self.active_method = args.active_method
        self.num_class = args.num_classes
        self.eps = 1e-8

    def calculate_scores(self, trainer, pool_set):
        raise NotImplementedError

    # -- helpers shared by the device selectors ---------------------------------------------------
    def valid_mask(self, pool_set):
        """u8 [n_img, S]: ids still listed in ``pool_set.suppix`` (``my_bvsb.py:41-46``)."""
        valid = np.zeros((len(pool_set.im_idx), self.num_superpixels), dtype=np.uint8)
        for k, key in enumerate(pool_set.im_idx):
            valid[k, pool_set.suppix[key[2]]] = 1
        return valid

    def gen_score_list_from_tensor(self, pool_set, scores_tensor):
        """(score, "img,lbl,spx", id) for every id still in the pool -- ``my_bvsb.py:29-48``."""
        scores = []
        host = scores_tensor.detach().cpu()
        for kdx, key in enumerate(pool_set.im_idx):
            path = ','.join(key)
            ids = pool_set.suppix[key[2]]
            scores.extend((s, path, i) for s, i in zip(host[kdx][ids].tolist(), ids))
        return scores

    def _save_scores(self, trainer, scores):
        fname = os.path.join(trainer.model_save_dir, "AL_record", "region_val_{}.json".format(trainer.selection_iter))
        os.makedirs(os.path.dirname(fname), exist_ok=True)
        with open(fname, "w") as f:
            json.dump(scores, f)

    def _region_cost(self, active_set, pool_set):
        """u8 [n_img, S] click cost per region, or None for unit cost
        (``dataloader/region_active_dataset.py:58-65``)."""
        args = self.args
        if not (getattr(args, 'fair_counting', False) and getattr(args, 'or_labeling', False)):
            return None
        label = active_set.trg_label_dataset
        rows = [label.id_to_index[key[2].split('/')[-1].split('.')[0]] for key in pool_set.im_idx]
        table = getattr(active_set, 'click_cost_table', None)
        if table is not None:                   # this package's RegionActiveDataset: popcounts computed once per run
            return np.ascontiguousarray(table()[rows])
        return np.ascontiguousarray(np.asarray(label.multi_hot_cls)[rows].sum(axis=2).astype(np.uint8))

    def _pool_valid(self, active_set, pool_set):
        mask = getattr(active_set, 'pool_valid_mask', None)
        return mask(self.num_superpixels) if mask is not None else self.valid_mask(pool_set)

    def select_next_batch(self, trainer, active_set, selection_count):
        pool_set = active_set.trg_pool_dataset
        if not hasattr(self, 'calculate_scores_tensor'):
            scores = self.calculate_scores(trainer, pool_set)
            if getattr(self.args, 'save_scores', False):
                self._save_scores(trainer, scores)
            active_set.expand_training_set(sorted(scores, reverse=True), selection_count, self.active_method)
            return
        scores_tensor = self.calculate_scores_tensor(trainer, pool_set)          # [n_img, S] on the device
        if getattr(self.args, 'save_scores', False):
            self._save_scores(trainer, self.gen_score_list_from_tensor(pool_set, scores_tensor))
        backend = self._backend(trainer)
        paths = [','.join(key) for key in pool_set.im_idx]
        from ..ops import path_ranks
        img_rank, img_of_rank = path_ranks(paths)
        dev = scores_tensor.device
        cost = self._region_cost(active_set, pool_set)
        from .engine import ShardPlan, current_rank_world, select_regions
        rank, world = current_rank_world()
        plan = ShardPlan(len(pool_set.im_idx), self.batch_size, rank, world)        # the sharding the scores were computed under
        n, simg, sid, ssc = select_regions(
            backend, plan, scores_tensor.contiguous(), torch.from_numpy(self._pool_valid(active_set, pool_set)).to(dev),
            torch.from_numpy(img_rank).to(dev), torch.from_numpy(img_of_rank).to(dev),
            None if cost is None else torch.from_numpy(cost).to(dev), int(selection_count),
            max_out=None if cost is not None and cost.min() == 0 else int(selection_count) + 1)
        # the consumed prefix as arrays that still read as the reference's list of (score, path, id) tuples
        from ..dataloader.region_active_dataset import ConsumedPrefix
        active_set.expand_training_set(ConsumedPrefix(ssc, simg, sid, pool_set.im_idx), selection_count, self.active_method)

    def _backend(self, trainer):
        b = getattr(self, 'backend', None)
        if b is None:
            from .engine import default_backend
            b = self.backend = default_backend(trainer.device)
        return b
