#!/usr/bin/env python
"""cProfile of the scan-only pool round of bench.py (2 975 pictures x 2 048 superpixels through RegionSelector.select_next_batch):
where does the host part go.   python tools/pool_round_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
bench.pool_round_bench(args, dev, 0, 1, False)          # warm (allocations, code paths)
pr = cProfile.Profile()
pr.enable()
out = bench.pool_round_bench(args, dev, 0, 1, False)
pr.disable()
print(out["seconds"], out["rank0_breakdown_s"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
