"""dev: one small predict_batch call taken apart -- G candidates (argv[1]), device draw, 40 calls: wall median, and (under
`rocprofv3 --kernel-trace --stats`) the kernels' device time per call."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from catgrasp_amd import synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else 'device'
dev = torch.device('cuda:0')
ob = synth.make_scene(1, 2048, seed=0)[0]
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
poses = list(synth.make_candidates(ob, G, np.random.default_rng(1)))
for _ in range(5):
    gp.predict_batch(data, poses, rng=mode)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); gp.predict_batch(data, poses, rng=mode); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f'G={G} rng={mode}: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms over 40 calls')
