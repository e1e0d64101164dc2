#!/usr/bin/env python
"""End-to-end demo on a synthetic clutter pile: for every object of the scene run the device pipeline
(occupancy -> NUNOCS + RANSAC -> cone candidates -> filterGraspPose -> affordance -> grasp-Q -> ranking) and print the
per-stage wall time -- first with the package's DEFAULT settings (exact f32, numpy's global stream for the resampling draw and the
2 x 10,000 RANSAC hypothesis draws: what a seeded reference run reproduces bit for bit), then with the fast non-reference draws
(ransac_sampling='fast', rng='device').  CATGRASP_AMD_PRECISION selects another arithmetic for both.  Weights are seeded random (the
reference's checkpoints are external downloads)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import engine, pipeline, synth                                     # noqa: E402
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter  # noqa: E402


def run(objs, g, gp, npred, scene_pts, K, rng, title):
    np.random.seed(0)
    per_ob, total, t0 = [], 0, time.perf_counter()
    # the loop over the scene's objects (run_grasp_simulation.py:188-329): with the reference's streams the next object's RANSAC hypothesis
    # draws are made ahead on a second thread while the device scores the current object -- same results as object-by-object calls
    outs = pipeline.evaluate_objects([{'ob_pts': ob['xyz'], 'ob_normals': ob['normal']} for ob in objs], scene_pts, K, g, gp, npred, timings=per_ob, rng=rng)
    for k, out in enumerate(outs):
        total += out['n_evaluated']
        print(f"object {k}: {out['n_evaluated']} candidates evaluated, {len(out['poses'])} survive, best P(T,G) = "
              f"{out['p_T_G'][0] if len(out['poses']) else float('nan'):.4f}")
    torch.cuda.synchronize()
    timings = {}
    for tm in per_ob[1:]:
        for name, v in tm.items():
            timings[name] = timings.get(name, 0.0) + v
    print(f'{title}: {total} candidates in {time.perf_counter() - t0:.2f} s wall (first object includes warm-up)')
    for name, v in timings.items():
        print(f'  {name:28s} {v * 1e3 / (len(objs) - 1):8.2f} ms / object')


def main():
    dev = torch.device('cuda:0')
    engine.set_precision(os.environ.get('CATGRASP_AMD_PRECISION', 'f32'))
    objs = synth.make_scene(8, 2500, seed=0)
    g = synth.make_gripper()
    g['finger_vertices'] = [g['vertices'][8:16], g['vertices'][16:24]]
    g['grip_dirs'] = [[0, -1, 0], [0, 1, 0]]
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
    sd_seg = synth.make_state_dict('seg', 6, 300, seed=1)
    scene_pts = np.concatenate([o['xyz'] for o in objs])
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    for title, sampling, rng in (("default (ransac_sampling='reference', rng='numpy')", 'reference', 'numpy'),
                                 ("fast draws (ransac_sampling='fast', rng='device')", 'fast', 'device')):
        npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=sd_seg, device=dev, ransac_sampling=sampling)
        run(objs, g, gp, npred, scene_pts, K, rng, title)


if __name__ == '__main__':
    main()
