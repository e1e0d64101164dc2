"""Where a small predict_batch call spends its time: cProfile of the host side over 200 calls + the wall clock, 1 and 16 poses, rng='device'."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter

dev = torch.device('cuda:0')
ob = synth.make_scene(1, 2048, seed=0)[0]
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
for G in (1, 16):
    poses = list(synth.make_candidates(ob, G, np.random.default_rng(1)))
    for _ in range(20):
        gp.predict_batch(data, poses, rng='device')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        gp.predict_batch(data, poses, rng='device')
    torch.cuda.synchronize()
    print(f'G={G}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call (back to back)')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        gp.predict_batch(data, poses, rng='device')
    pr.disable()
    st = pstats.Stats(pr).sort_stats('tottime')
    st.print_stats(18)
