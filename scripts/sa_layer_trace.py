#!/usr/bin/env python
"""The set-abstraction layer as the drop-in issues it, 30 times at one 20,000-point cloud and 30 times at 16, under
`rocprofv3 --kernel-trace --stats` -> profiles/r4_sa_layer_kernel_stats.csv (which kernels a layer is, and their durations)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import primitives   # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
N, S, K, R = 20000, 1024, 32, 0.02
sa = primitives.SetAbstractionWeights([(np.random.default_rng(0).normal(0, 0.2, (64, 9)), np.zeros(64), None), (np.random.default_rng(1).normal(0, 0.1, (64, 64)), np.zeros(64), None),
                                       (np.random.default_rng(2).normal(0, 0.1, (128, 64)), np.zeros(128), None)], 9, dev)
for B in (1, 16):
    pts = (torch.rand(B, N, 3, device=dev, generator=g) * 0.1).contiguous(); feat = torch.randn(B, N, 6, device=dev, generator=g)
    start = torch.zeros(B, dtype=torch.long, device=dev)
    for _ in range(30):
        _, nx = primitives.farthest_point_sample(pts, S, start=start, return_xyz=True)
        ix = primitives.query_ball_point(R, K, pts, nx)
        primitives.group_mlp_max(pts, feat, nx, ix, sa, check_indices=False)
    torch.cuda.synchronize()
