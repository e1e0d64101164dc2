"""Latency of GraspPredicter.predict_batch through the reference entry point at small candidate counts (C1 = 256 candidates on a
2,048-point cloud), numpy's stream, f32: median of 20 calls after 3 warm-ups."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from catgrasp_amd import synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
dev = torch.device('cuda:0')
ob = synth.make_scene(1, 2048, seed=0)[0]
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
out = []
for G in (1, 16, 256, 1024, 4096):
    poses = list(synth.make_candidates(ob, G, np.random.default_rng(1)))
    for mode in ('numpy', 'device'):
        ts = []
        for i in range(23):
            np.random.seed(i); torch.cuda.synchronize(); t0 = time.perf_counter(); gp.predict_batch(data, poses, rng=mode); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[3:])
        out.append({'candidates': G, 'rng': mode, 'median_ms': round(ts[len(ts) // 2] * 1e3, 3), 'min_ms': round(ts[0] * 1e3, 3),
                    'matrix_time_ms_at_12.1us_per_candidate': round(G * 12.1e-3, 3)})
        print(out[-1], flush=True)
print(json.dumps(out))
