#!/usr/bin/env python
"""filterGraspPose on the device, C3 call shapes (cone poses x [I], adjust off; canonical grasps x 12 nut symmetries, adjust on) against
the toy (36 / 48 triangles) and the subdivided (9,216 / 12,288 triangles) gripper: HIP-event time per call, evaluations/s, and the
broad-phase (grid) kernel's codes / poses / nudges against the exhaustive kernel's (must be identical)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import my_cpp, synth, transforms  # noqa: E402

dev = torch.device('cuda:0')
objs = synth.make_scene(8, 2500, 0); g = synth.make_gripper(); bg = synth.background_points(objs, 0, g['diameter'])
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
base = synth.make_candidates(objs[0], n, rng, g['hand_depth'], g['init_bite'])
cone = torch.from_numpy(base.astype(np.float32).reshape(-1, 16)).to(dev)
nocs = objs[0]['pose'] @ np.diag([0.02, 0.02, 0.02, 1.0])
sym12 = torch.from_numpy(np.stack(transforms.get_symmetry_tfs('nut')).astype(np.float32).reshape(-1, 16)).to(dev)
can = torch.from_numpy((np.linalg.inv(nocs) @ base[:(n + 11) // 12]).astype(np.float32).reshape(-1, 16)).to(dev)
sym1 = torch.eye(4, device=dev).reshape(1, 16); I4 = np.eye(4, dtype=np.float32)
for sub in (0, 4):
    V, F = synth.subdivide(g['vertices'], g['faces'], sub); Ve, Fe = synth.subdivide(g['enclosed_vertices'], g['enclosed_faces'], sub)
    ref = {}
    for accel in (False, True):
        sc = my_cpp.GripperScene(V, F, Ve, Fe, objs[0]['xyz'], bg, 0.0005, dev, accel=accel, cache=False)
        for name, (P, S, npose, adj) in {'cone x [I], adjust off': (cone, sym1, I4, False), 'canonical x 12, adjust on': (can, sym12, nocs.astype(np.float32), True)}.items():
            if not accel and sub > 0:
                P = P[:max(1, len(P) // 25)]                                # the exhaustive kernel on 9k triangles: a sample is enough
            call = lambda: my_cpp.filter_on_device(sc, P, S, npose, I4, I4, I4, g['gripper_in_grasp'], True, False, adj)
            for _ in range(2):
                out = call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = call()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            E = out[0].numel()
            tag = ''
            if not accel:
                ref[name] = [o.cpu() for o in out]
            else:
                r = ref[name]; k = r[0].numel()
                same = all(torch.equal(o.cpu().reshape(E, -1)[:k], q.reshape(k, -1)) for o, q in zip(out, r))
                tag = f' identical to the exhaustive kernel on the first {k}: {same}'
                assert same
            print(f'tris={len(F)}/{len(Fe)} grid={accel} {name}: {ms:.3f} ms per {E} evaluations ({E / ms * 1e3 / 1e6:.2f} M/s) '
                  f'codes={torch.bincount(out[0].long(), minlength=5).tolist()}{tag}', flush=True)
