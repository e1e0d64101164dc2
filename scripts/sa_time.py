"""dev: average duration of the fused set-abstraction launch (9-64-64-128, N = 20,000, S = 1,024, K = 32) for 1 and 16 clouds, HIP events
around 50 back-to-back launches (CATGRASP_AMD_LIB selects the build)."""
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
for Bb in (1, 16, 48):
    pts = (torch.rand(Bb, N, 3, device=dev, generator=g) * 0.1).contiguous(); feat = torch.randn(Bb, N, 6, device=dev, generator=g)
    new = pts[:, :S].contiguous()
    idx = primitives.query_ball_point(0.02, K, pts, new); idx = torch.where(idx >= N, torch.zeros_like(idx), idx)
    for _ in range(5):
        out = primitives.group_mlp_max(pts, feat, new, idx, sa, check_indices=False)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            out = primitives.group_mlp_max(pts, feat, new, idx, sa, check_indices=False)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    o = out[0] if isinstance(out, tuple) else out
    print(f'clouds {Bb:2d}: {best:7.2f} us  checksum {float(o.double().sum()):.6f}', flush=True)
