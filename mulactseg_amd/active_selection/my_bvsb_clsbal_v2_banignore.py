"""BvSB + region-level class balancing (Cai et al., CVPR 2021) + ban of "undefined"-dominated regions --
reference ``active_selection/my_bvsb_clsbal_v2_banignore.py``: after normalisation and ban, every region is
weighted by ``exp(-share of regions with the same dominant class)`` (:66-74; the share counts ALL
``n_img * nseg`` rows, absent regions included with dominant class 0)."""
import numpy as np
import torch

from . import my_bvsb_banignore


class RegionSelector(my_bvsb_banignore.RegionSelector):
    extra_channels = 1
    ban_ignore = True
    class_balance = True

    def _class_weight(self, backend, dominant, C):
        counts = backend.dominant_hist(dominant.contiguous(), C).cpu().numpy().astype(np.int64)
        # torch: int64 / int64 -> float32 true division; exp in f32 (here: f64 exp rounded once to f32)
        dist = counts.astype(np.float32) / np.float32(counts.sum())
        self.est_label_dist = dist
        w = np.exp(-dist.astype(np.float64)).astype(np.float32)
        self.cls_weight = torch.from_numpy(w).to(dominant.device)
        return self.cls_weight
