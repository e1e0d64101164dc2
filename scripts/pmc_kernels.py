"""The PointNet++ / collision kernels BASELINE.json's north_star names, alone, for a `rocprofv3 --pmc` pass:
sa_group_mlp_max_kernel (16 clouds and one cloud), farthest_point_sample_kernel (one 20k cloud and the 8 clouds of C3 in one launch),
filter_grasp_pose_kernel (C3 call shape, broad-phase grid).  Few launches each: counter passes serialise kernels.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d <dir> -- python scripts/pmc_kernels.py
    rocprofv3 --pmc FETCH_SIZE ...   /   rocprofv3 --pmc WRITE_SIZE ...          (separate passes)
    python scripts/pmc_summary.py <dir> profiles/r3_pmc_sq_northstar_kernels.csv
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import my_cpp, primitives, synth   # noqa: E402

dev = torch.device('cuda:0')
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator(device=dev); g.manual_seed(0)
N, S, K = 20000, 1024, 32
sa = primitives.SetAbstractionWeights([(np.random.default_rng(0).normal(0, 0.2, (64, 9)), np.zeros(64), None), (np.random.default_rng(1).normal(0, 0.1, (64, 64)), np.zeros(64), None),
                                       (np.random.default_rng(2).normal(0, 0.1, (128, 64)), np.zeros(128), None)], 9, dev)
for Bb in (16, 1):
    pts = (torch.rand(Bb, N, 3, device=dev, generator=g) * 0.1).contiguous(); feat = torch.randn(Bb, N, 6, device=dev, generator=g)
    new = pts[:, :S].contiguous()
    idx = primitives.query_ball_point(0.02, K, pts, new); idx = torch.where(idx >= N, torch.zeros_like(idx), idx)
    for _ in range(ITERS):
        primitives.group_mlp_max(pts, feat, new, idx, sa)
    torch.cuda.synchronize()
for Bb in (1, 8):
    pts = (torch.rand(Bb, N, 3, device=dev, generator=g) * 0.1).contiguous()
    for _ in range(max(2, ITERS // 2)):
        primitives.farthest_point_sample(pts, S, start=torch.zeros(Bb, dtype=torch.long, device=dev))
    torch.cuda.synchronize()
objs = synth.make_scene(8, 2500, seed=0)
gr = synth.make_gripper()
bg = synth.background_points(objs, 0, gr['diameter'])
sc = my_cpp.GripperScene(gr['vertices'], gr['faces'], gr['enclosed_vertices'], gr['enclosed_faces'], objs[0]['xyz'], bg, 0.0005, dev)
Pc = torch.from_numpy(synth.make_candidates(objs[0], 50000, np.random.default_rng(0), gr['hand_depth'], gr['init_bite']).astype(np.float32).reshape(-1, 16)).to(dev)
sym = torch.eye(4, device=dev).reshape(1, 16)
I4 = np.eye(4, dtype=np.float32)
for adjust in (False, True):
    for _ in range(ITERS):
        my_cpp.filter_on_device(sc, Pc, sym, I4, I4, I4, I4, gr['gripper_in_grasp'], True, False, adjust)
torch.cuda.synchronize()
print('pmc_kernels done')
