"""Duration of the fused set-abstraction kernel against the number of clouds (rocprofv3 kernel trace): slope = steady-state cost per tile,
intercept = fixed cost of a launch (workgroup dispatch, weight staging, tail)."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import primitives   # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
N, S, K = 20000, 1024, 32
sa = primitives.SetAbstractionWeights([(np.random.default_rng(0).normal(0, 0.2, (64, 9)), np.zeros(64), None), (np.random.default_rng(1).normal(0, 0.1, (64, 64)), np.zeros(64), None),
                                       (np.random.default_rng(2).normal(0, 0.1, (128, 64)), np.zeros(128), None)], 9, dev)
for Bb in (1, 2, 4, 8, 16, 24, 32, 48):
    pts = (torch.rand(Bb, N, 3, device=dev, generator=g) * 0.1).contiguous(); feat = torch.randn(Bb, N, 6, device=dev, generator=g)
    new = pts[:, :S].contiguous()
    idx = primitives.query_ball_point(0.02, K, pts, new); idx = torch.where(idx >= N, torch.zeros_like(idx), idx)
    for _ in range(6):
        primitives.group_mlp_max(pts, feat, new, idx, sa, check_indices=False)
    torch.cuda.synchronize()
print('done')
