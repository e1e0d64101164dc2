"""dev: where the host time of a small predict_batch call goes (cProfile, 200 calls of 16 candidates, device draw and numpy stream)."""
import cProfile, pstats, sys, io
import numpy as np, torch
sys.path.insert(0, '.')
from catgrasp_amd import synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
dev = torch.device('cuda:0')
ob = synth.make_scene(1, 2048, seed=0)[0]
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
poses = list(synth.make_candidates(ob, 16, np.random.default_rng(1)))
for mode in ('device', 'numpy'):
    for _ in range(5):
        gp.predict_batch(data, poses, rng=mode)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        gp.predict_batch(data, poses, rng=mode)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(mode); print(s.getvalue()[:5200])
