#!/usr/bin/env python
"""filter_grasp_pose kernels alone for rocprofv3 passes: the C3 call shapes (cone poses x [I], adjust off; canonical grasps x 12 nut
symmetries, adjust on) on the subdivided 9,216 / 12,288-triangle gripper, 50,000 evaluations per launch.  Prints the grid kernel's
own work counters (voxel keys read, grid cells looked up, pairs tested) for the cache-level byte count.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
        --kernel-include-regex 'filter_grasp_pose|compose_grasp' --output-format csv -d <dir> -- python scripts/pmc_filter.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import my_cpp, synth, transforms  # noqa: E402

dev = torch.device('cuda:0')
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
objs = synth.make_scene(8, 2500, 0); g = synth.make_gripper(); bg = synth.background_points(objs, 0, g['diameter'])
rng = np.random.default_rng(1)
n = 50000
base = synth.make_candidates(objs[0], n, rng, g['hand_depth'], g['init_bite'])
cone = torch.from_numpy(base.astype(np.float32).reshape(-1, 16)).to(dev)
nocs = objs[0]['pose'] @ np.diag([0.02, 0.02, 0.02, 1.0])
sym12 = torch.from_numpy(np.stack(transforms.get_symmetry_tfs('nut')).astype(np.float32).reshape(-1, 16)).to(dev)
can = torch.from_numpy((np.linalg.inv(nocs) @ base[:(n + 11) // 12]).astype(np.float32).reshape(-1, 16)).to(dev)
sym1 = torch.eye(4, device=dev).reshape(1, 16); I4 = np.eye(4, dtype=np.float32)
V, F = synth.subdivide(g['vertices'], g['faces'], 4); Ve, Fe = synth.subdivide(g['enclosed_vertices'], g['enclosed_faces'], 4)
sc = my_cpp.GripperScene(V, F, Ve, Fe, objs[0]['xyz'], bg, 0.0005, dev)
for name, (P, S, npose, adj) in {'cone x [I], adjust off': (cone, sym1, I4, False), 'canonical x 12, adjust on': (can, sym12, nocs.astype(np.float32), True)}.items():
    stats = torch.zeros(3, dtype=torch.int64, device=dev)
    for it in range(ITERS):
        out = my_cpp.filter_on_device(sc, P, S, npose, I4, I4, I4, g['gripper_in_grasp'], True, False, adj, work_stats=stats if it == 0 else None)
    torch.cuda.synchronize()
    w = stats.cpu().numpy()
    E = out[0].numel()
    print(f'{name}: {E} evaluations, voxel keys read {w[0]}, cells looked up {w[1]}, pairs tested {w[2]} -> cache-level bytes '
          f'{8 * w[0] + 8 * w[1] + (48 + 4) * w[2] + 130 * E}', flush=True)
print('pmc_filter done')
