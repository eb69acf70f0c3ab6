"""Random region selector -- reference ``active_selection/my_random.py`` (initial round).

``calculate_scores`` is the reference's: one ``random.random()`` per pool region, in pool order.  ``select_next_batch``
consumes the SAME draws in the same order (the global RNG ends in the same state) but orders them with one numpy
``lexsort`` on (draw, path rank, id) -- the order of ``sorted(tuples, reverse=True)`` (``base.py:37``) -- instead of
building and sorting ~6 M Python tuples for the Cityscapes pool, and hands ``expand_training_set`` only the prefix
the budget consumes."""
import random

import numpy as np

from . import base


class RegionSelector(base.RegionSelector):
    def calculate_scores(self, trainer, pool_set):
        scores = []
        for key in pool_set.im_idx:
            path = ",".join(key)
            for suppix_id in pool_set.suppix[key[2]]:
                scores.append((random.random(), path, suppix_id))
        return scores

    def select_next_batch(self, trainer, active_set, selection_count):
        pool_set = active_set.trg_pool_dataset
        if getattr(self.args, 'save_scores', False) or not hasattr(active_set, 'click_cost_table'):
            return super().select_next_batch(trainer, active_set, selection_count)
        paths = [','.join(key) for key in pool_set.im_idx]
        lens = np.fromiter((len(pool_set.suppix[key[2]]) for key in pool_set.im_idx), dtype=np.int64, count=len(paths))
        n = int(lens.sum())
        rnd = random.random
        draws = np.fromiter((rnd() for _ in range(n)), dtype=np.float64, count=n)           # pool order, as calculate_scores
        img = np.repeat(np.arange(len(paths)), lens)
        ids = np.fromiter((i for key in pool_set.im_idx for i in pool_set.suppix[key[2]]), dtype=np.int64, count=n)
        from ..ops import path_ranks
        rank, _ = path_ranks(paths)
        order = np.lexsort((-ids, -rank[img].astype(np.int64), -draws))                      # descending tuples
        cost_tab = active_set.click_cost_table()
        if cost_tab is None:
            m = min(n, int(selection_count) + 1)
        else:
            rows = np.fromiter((active_set._image_index(key[2]) for key in pool_set.im_idx), dtype=np.int64, count=len(paths))
            cum = np.cumsum(cost_tab[rows[img[order]], ids[order]].astype(np.int64))
            over = np.nonzero(cum > selection_count)[0]
            m = n if len(over) == 0 else int(over[0]) + 1
        head = order[:m]
        consumed = [(d, paths[i], r) for d, i, r in zip(draws[head].tolist(), img[head].tolist(), ids[head].tolist())]
        active_set.expand_training_set(consumed, selection_count, self.active_method)
