"""Kernel time of the LDS-tile fused set-abstraction layer (csrc/sa_tile.hip) alone on PointNet++ layer shapes, HIP events around 20
back-to-back launches: us per launch, TFLOP/s on the algorithmic flops (2 K sum cin cout per neighbourhood), fraction of the f32 MFMA
peak.  CATGRASP_AMD_LIB selects an ablation build (scripts/build_abl.sh).    python scripts/sa_tile_time.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import pointnet2 as p2            # noqa: E402
from catgrasp_amd import primitives as prim        # noqa: E402

PEAK = 157.3e12
dev = torch.device('cuda:0')
# name, N (points the layer gathers from), S, K, D, mlp
SHAPES = [('ssg_sa2', 512, 128, 64, 128, [128, 128, 256]),
          ('ssg_sa1', 20000, 512, 32, 3, [64, 64, 128]),
          ('msg_sa1_s0', 20000, 512, 16, 3, [32, 32, 64]),
          ('first_level_256', 20000, 512, 32, 3, [64, 128, 256]),
          ('first_level_4_layers', 20000, 512, 32, 3, [32, 32, 64, 64]),
          ('msg_sa1_s2', 20000, 512, 128, 3, [64, 96, 128]),
          ('msg_sa2_s0', 512, 128, 32, 320, [64, 64, 128]),
          ('msg_sa2_s1', 512, 128, 64, 320, [128, 128, 256]),
          ('msg_sa2_s2', 512, 128, 128, 320, [128, 128, 256])]


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


rows = []
for name, N, S, K, D, mlp in SHAPES:
    for B in (1, 8, 16, 64):
        g = torch.Generator(device=dev); g.manual_seed(B)
        xyz = torch.rand(B, N, 3, device=dev, generator=g)
        pts = torch.randn(B, N, D, device=dev, generator=g)
        new_xyz = xyz[:, :S].contiguous()
        idx = torch.randint(0, N, (B, S, K), device=dev, generator=g)
        torch.manual_seed(0)
        sa = p2.PointNetSetAbstraction(S, 0.2, K, 3 + D, mlp).to(dev).eval()
        W = prim.SetAbstractionWeights(p2._sa_layers_from_state(sa.state_dict(), 'mlp_', len(mlp)), 3 + D, dev, kind='tile')
        out = torch.empty(B, S, mlp[-1], device=dev)
        t = timed(lambda: prim.group_mlp_max(xyz, pts, new_xyz, idx, W, check_indices=False, channels_last=True, out=out))
        fl, prev = 0, 3 + D
        for c in mlp:
            fl += 2 * prev * c; prev = c
        fl *= B * S * K
        rows.append({'shape': name, 'clouds': B, 'S': S, 'K': K, 'cin': 3 + D, 'mlp': mlp, 'us': round(t * 1e6, 2), 'tflops': round(fl / t / 1e12, 2),
                     'frac_of_f32_mfma_peak': round(fl / t / PEAK, 4)})
        if 3 + D <= 16 and max(mlp) <= 256:      # a first-level shape: the kernels of setabstraction.hip (register-resident, or its LDS-strip fallback) beside it
            Wr = prim.SetAbstractionWeights(p2._sa_layers_from_state(sa.state_dict(), 'mlp_', len(mlp)), 3 + D, dev, kind='reg')
            tr = timed(lambda: prim.group_mlp_max(xyz, pts, new_xyz, idx, Wr, check_indices=False, channels_last=True, out=out))
            rows[-1]['us_setabstraction_hip_kernels'] = round(tr * 1e6, 2)
        print(rows[-1], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump({'lib': os.environ.get('CATGRASP_AMD_LIB', 'default'), 'rows': rows}, f, indent=1)
