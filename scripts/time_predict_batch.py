"""Wall-clock of GraspPredicter.predict_batch (the reference entry point) on 50k poses: rng='numpy' (reference-exact stream) and
rng='device', f32 and f16x3 -- the api block of bench.py without the rest of the bench."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import engine, synth                                            # noqa: E402
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter              # noqa: E402

dev = torch.device('cuda:0')
ob = synth.make_scene(8, 2500, seed=0)[0]
rng = np.random.default_rng(5)
base = synth.make_candidates(ob, 2000, rng)
n = 50000
poses = list(base[rng.integers(0, len(base), n)])
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
out = []
for prec in ('f32', 'f16x3'):
    with engine.precision(prec):
        gp.predict_batch(data, poses[:2000], rng='device')
        for mode in ('device', 'numpy', 'numpy'):
            np.random.seed(0)
            torch.cuda.synchronize(); t0 = time.perf_counter(); gp.predict_batch(data, poses, rng=mode); torch.cuda.synchronize()
            t = time.perf_counter() - t0
            out.append({'precision': prec, 'rng': mode, 'wall_s': round(t, 4), 'candidates_per_s': round(n / t, 1)})
print(json.dumps(out))
