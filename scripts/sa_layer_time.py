#!/usr/bin/env python
"""One PointNet++ set-abstraction layer end to end -- farthest_point_sample (which also writes new_xyz = index_points(xyz, fps_idx)) ->
query_ball_point -> fused group-MLP-max (pointnet2.py:54-129 + the Conv2d/BN/ReLU/max that consumes it) -- at N = 20,000, S = 1,024,
K = 32, 9-64-64-128, for 1 / 8 / 16 clouds per call, on a uniform volume and on a surface cloud (what a depth camera gives): HIP-event
time per stage (back-to-back launches of the stage alone; `index_points` is listed for reference, the layer no longer issues it) and of
the whole layer as the drop-in issues it (pointnet2.PointNetSetAbstraction.forward).  -> profiles/r4_sa_layer.json (argv[1])."""
import json
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
FLOP_PER_NEIGHBOURHOOD = 2 * K * (9 * 64 + 64 * 64 + 64 * 128)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def surface(B):
    """a corrugated tube, points in random index order"""
    rng = np.random.default_rng(3)
    t = rng.random((B, N, 2)); r = 0.03 + 0.01 * np.sin(12 * t[..., 0] * np.pi)
    return torch.from_numpy(np.stack([r * np.cos(2 * np.pi * t[..., 0]), r * np.sin(2 * np.pi * t[..., 0]), 0.08 * t[..., 1]], -1).astype(np.float32)).to(dev)


rows = []
for cloud, B in (('uniform volume', 1), ('uniform volume', 8), ('uniform volume', 16), ('surface', 1), ('surface', 8), ('surface', 16)):
    pts = (torch.rand(B, N, 3, device=dev, generator=g) * 0.1).contiguous() if cloud == 'uniform volume' else surface(B)
    feat = torch.randn(B, N, 6, device=dev, generator=g)
    start = torch.zeros(B, dtype=torch.long, device=dev)
    fps = primitives.farthest_point_sample(pts, S, start=start)
    new = primitives.index_points(pts, fps).contiguous()
    idx = primitives.query_ball_point(R, K, pts, new)

    def layer():
        _, nx = primitives.farthest_point_sample(pts, S, start=start, return_xyz=True)      # the sampled points come out of the FPS launch
        ix = primitives.query_ball_point(R, K, pts, nx)
        return primitives.group_mlp_max(pts, feat, nx, ix, sa, check_indices=False)
    t = {'fps': timed(lambda: primitives.farthest_point_sample(pts, S, start=start), 5), 'index_points': timed(lambda: primitives.index_points(pts, fps)),
         'query_ball_point': timed(lambda: primitives.query_ball_point(R, K, pts, new)),
         'group_mlp_max': timed(lambda: primitives.group_mlp_max(pts, feat, new, idx, sa, check_indices=False), 50), 'layer': timed(layer, 5)}
    row = {'cloud': cloud, 'clouds': B, 'ms': {k: round(v, 4) for k, v in t.items()}, 'fps_us_per_round': round(t['fps'] / S * 1e3, 3),
           'fps_share_of_layer': round(t['fps'] / t['layer'], 3),
           'group_mlp_max_tflops': round(B * S * FLOP_PER_NEIGHBOURHOOD / (t['group_mlp_max'] * 1e-3) / 1e12, 2),
           'group_mlp_max_frac_of_157.3': round(B * S * FLOP_PER_NEIGHBOURHOOD / (t['group_mlp_max'] * 1e-3) / 1e12 / 157.3, 3)}
    rows.append(row); print(row, flush=True)
out = {'what': __doc__.strip(), 'N': N, 'S': S, 'K': K, 'radius': R, 'layers': '9-64-64-128', 'rows': rows}
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(out, f, indent=1)
