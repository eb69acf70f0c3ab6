#!/usr/bin/env python
"""Host cost of RegionActiveDataset.expand_training_set at pool scale (2 975 pictures x 2 048 superpixels, 100 000 clicks, fair
counting) on synthetic lists -- CPU only.   python tools/expand_profile.py [--profile]"""
import os
import sys
import tempfile
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd.dataloader.region_active_dataset import ConsumedPrefix, RegionActiveDataset     # noqa: E402


class Pool:
    def __init__(self, n, S):
        names = ["city_%05d" % i for i in range(n)]
        self.im_idx = [["leftImg8bit/%s.png" % m, "gtFine/%s.png" % m, "superpixel/%s.pkl" % m] for m in names]
        self.suppix = {k[2]: list(range(S)) for k in self.im_idx}
        self.isselected = np.zeros((n, S), dtype=np.uint8)
        self.n, self.S = n, S
        self.suppix_ascending = os.environ.get('ASC', '1') == '1'

    def initial_valid_table(self):
        return np.ones((self.n, self.S), dtype=np.uint8)


def main():
    n, S, C, clicks = 2975, 2048, 20, 100000
    rs = np.random.RandomState(3)
    best = []
    for rep in range(4):
        pool = Pool(n, S)
        labels = types.SimpleNamespace(im_idx=[], suppix={}, multi_hot_cls=(rs.rand(n, S, C) < 0.07).astype(np.uint8),
                                       id_to_index={k[2].split('/')[-1].split('.')[0]: i for i, k in enumerate(pool.im_idx)})
        labels.multi_hot_cls[:, :, 0] |= (labels.multi_hot_cls.sum(2) == 0)
        tmp = tempfile.mkdtemp()
        args = types.SimpleNamespace(fair_counting=True, or_labeling=True, model_save_dir=tmp, finetune_itrs=1, wandb=None)
        act = RegionActiveDataset(args, pool, labels)
        act.selection_iter = 1
        act.click_cost_table()
        act.pool_valid_mask(S)
        m = 80000
        flat = rs.choice(n * S, size=m, replace=False)
        sr = ConsumedPrefix(np.sort(rs.rand(m).astype(np.float32))[::-1], flat // S, flat % S, pool.im_idx)
        if '--profile' in sys.argv and rep == int(os.environ.get('REP', '3')):
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            act.expand_training_set(sr, clicks, 'pixbal')
            pr.disable()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
        else:
            t0 = time.perf_counter()
            k = act.expand_training_set(sr, clicks, 'pixbal')
            t1 = time.perf_counter()
            act.wait_for_writes()
            t2 = time.perf_counter()
            best.append(t1 - t0)
            t3 = time.perf_counter()
            act.dump_datalist()
            t4 = time.perf_counter()
            print("selected %d regions: expand %.4f s, pickle still writing %.4f s; label lists %d; dump_datalist %.3f s" % (k, t1 - t0, t2 - t1, len(labels.suppix), t4 - t3))
    print("best %.4f s" % min(best))


if __name__ == "__main__":
    main()
