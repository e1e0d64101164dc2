import sys, time, gc
import numpy as np, torch
sys.path.insert(0, '.')
from catgrasp_amd import engine, synth, ops
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
from catgrasp_amd.workload import SceneBatch
from catgrasp_amd import distributed as cgd
dev = torch.device('cuda:0')
sd = synth.make_state_dict('cls', 6, 10, seed=0)
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=dev)
npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'with_batch'
ob = synth.make_scene(8, 2500, seed=0)[0]
rng = np.random.default_rng(5)
base = synth.make_candidates(ob, 2000, rng)
n = 50000
poses = list(base[rng.integers(0, len(base), n)])
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
def t_api(tag, rngmode):
    gp.predict_batch(data, poses[:2000], rng=rngmode)
    for i in range(4):
        np.random.seed(0); torch.cuda.synchronize(); t0 = time.perf_counter(); gp.predict_batch(data, poses, rng=rngmode); torch.cuda.synchronize()
        print(tag, rngmode, i, round(time.perf_counter() - t0, 4), flush=True)
if mode == 'with_batch':
    batch = SceneBatch(dev, {'nut': gp}, {'nut': npred}, kind='nut', n_objects=8, pts_per_object=2500, per_replica=50000, replicas=1)
    with torch.no_grad():
        for _ in range(8):
            out = cgd.score_sharded(batch.score_slice, 50000)
    torch.cuda.synchronize()
    t_api('after-steps', 'device'); t_api('after-steps', 'numpy')
    gc.collect(); gc.freeze()
    t_api('gc-frozen', 'device'); t_api('gc-frozen', 'numpy')
else:
    t_api('plain', 'device'); t_api('plain', 'numpy')
