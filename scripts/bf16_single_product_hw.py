#!/usr/bin/env python
"""BASELINE.json configs[4] names a "bf16 MLP path with MFMA"; the product runs C5 as bf16x3 (three bf16 MFMAs per product block).
What does ONE bf16 MFMA per block cost in accuracy ON HARDWARE?  A lower bound that needs no new kernel: feed the shipped bf16x3
kernels weights whose low pieces are zero.  They then compute x . bf16(w) with the activations still carried as hi + lo pieces --
i.e. only the WEIGHT rounding of a plain-bf16 path (a true single-MFMA path also rounds every activation to bf16: its error is
larger, about sqrt(2) x in the emulation of scripts/bf16x3_sim.py).  Logits / probabilities / p_G of PointNetCls on 256 candidates of
the C3 scene against the exact-f32 kernels and against the float64 oracle evaluation.  -> JSON on stdout (argv[1]: output file)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import engine, folding, ops, synth, transforms      # noqa: E402
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter   # noqa: E402

dev = torch.device('cuda:0')
ob = synth.make_scene(8, 2500, seed=0)[0]
P = synth.make_candidates(ob, 256, np.random.default_rng(1))
report = {'what': __doc__.strip().split('\n')[0], 'candidates': len(P), 'networks': []}
for seed in (0, 2, 4):                 # the seeded random-init networks of bench.py (gain 1.6)
    sd = synth.make_state_dict('cls', 6, 10, seed=seed)
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=dev)
    cloud = gp.upload_cloud({'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']})
    ids = transforms.draw_ids_device(cloud.n, 2048, len(P), dev, seed=7)
    pinv = torch.from_numpy(transforms.pose_inverse_rows(P, cloud.center)).to(dev)
    x = ops.build_grasp_input(cloud.xyz, cloud.normal, ids, pinv, gp._mean, gp._inv_std)
    with torch.no_grad():
        with engine.precision('f32'):
            l32 = engine.cls_forward(gp._W, x)[0].double().cpu().numpy()
        with engine.precision('bf16x3'):
            l3 = engine.cls_forward(gp._W, x)[0].double().cpu().numpy()
        real_split = folding.bf16_split
        folding.bf16_split = lambda w: (real_split(w)[0], np.zeros_like(real_split(w)[1]))          # weights: hi piece only
        try:
            W1 = folding.prepare_cls(sd, dev)
        finally:
            folding.bf16_split = real_split
        with engine.precision('bf16x3'):
            l1 = engine.cls_forward(W1, x)[0].double().cpu().numpy()
    # float64 evaluation of the same network (oracle module; this is a dev script, not a product path)
    from oracle import pointnet_ref as oref
    l64 = oref.pointnet_cls_forward(sd, x.double().cpu(), dtype=torch.float64)[0].numpy()

    def sm(l):
        e = np.exp(l - l.max(1, keepdims=True)); return e / e.sum(1, keepdims=True)
    row = {'seed': seed}
    for name, l in (('f32', l32), ('bf16x3', l3), ('bf16_weights_only_single_product', l1)):
        row[name] = {'max_abs_logit_err_vs_f64': float(np.abs(l - l64).max()), 'max_abs_prob_err_vs_f64': float(np.abs(sm(l) - sm(l64)).max()),
                     'max_abs_pG_err_vs_f64': float(np.abs((sm(l) * np.arange(10)).sum(1) / 10 - (sm(l64) * np.arange(10)).sum(1) / 10).max())}
    report['networks'].append(row); print(row, file=sys.stderr, flush=True)
report['bar'] = 1e-4
print(json.dumps(report, indent=1))
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(report, f, indent=1)
