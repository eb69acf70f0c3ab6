"""Random region selector -- reference ``active_selection/my_random.py`` (initial round)."""
import random

from . import base


class RegionSelector(base.RegionSelector):
    def calculate_scores(self, trainer, pool_set):
        scores = []
        for key in pool_set.im_idx:
            path = ",".join(key)
            for suppix_id in pool_set.suppix[key[2]]:
                scores.append((random.random(), path, suppix_id))
        return scores
