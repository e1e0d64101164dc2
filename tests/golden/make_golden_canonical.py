"""Golden FILES in the reference's own on-disk formats, written with the REAL reference classes:
  tests/golden/canonical_golden.pkl       -- `{class}_canonical.pkl` as make_canonical.py:153-164 writes it (gzip pickle of a dict with numpy
                                             arrays and a numpy object array of dexnet.grasping.grasp.ParallelJawPtGrasp3D whose contacts
                                             are dexnet.grasping.contacts.Contact3D), read at run_grasp_simulation.py:706-707;
  tests/golden/complete_grasp_golden.pkl  -- a `*_complete_grasp.pkl` grasp list (generate_grasp.py / make_canonical.py:108-110).
Build container only (imports /root/reference under the inert stubs of make_golden_sampler.py).

    python tests/golden/make_golden_canonical.py
"""
import gzip
import os
import pickle
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# the stub finder + the import of the real dexnet.grasping modules live in make_golden_sampler.py; reuse its module state
ns = runpy.run_path(os.path.join(HERE, 'make_golden_sampler.py'), run_name='not_main')
ref_sampler = ns['ref_sampler']
from dexnet.grasping.contacts import Contact3D           # noqa: E402  (real class, importable under the stubs)
from dexnet.grasping.grasp import ParallelJawPtGrasp3D   # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from catgrasp_amd import synth                            # noqa: E402

rng = np.random.default_rng(2024)


def contact(p, d):
    c = Contact3D.__new__(Contact3D)       # as pickled: the constructor needs a GraspableObject3D, the saved state does not
    c.graspable_, c.point_, c.in_direction_, c.friction_cone_, c.normal_, c.surface_info_ = object(), p, d, None, -d, None
    return c


grasps = []
for i in range(9):
    T = np.eye(4); T[:3, :3] = synth.random_rotation(rng); T[:3, 3] = rng.normal(0, 0.01, 3)
    c1p, c2p = T[:3, 3] + 0.01 * T[:3, 1], T[:3, 3] - 0.01 * T[:3, 1]
    g = ParallelJawPtGrasp3D(grasp_pose=T, c1=contact(c1p, -T[:3, 1]), c2=contact(c2p, T[:3, 1]), friction_score=float(rng.uniform()),
                             canny_quality=float(rng.uniform()), perturbation_score=float(rng.uniform(0.5, 1.0)), grasp_id=i)
    grasps.append(g)
assert all(g.c1.graspable_ is None for g in grasps)      # set_contacts (grasp.py:149-156) drops the graspable before saving

pts, nrm = synth.nut_surface(64, rng)
canonical = {'obj_files': ['/data/object_models/nut_0.obj', '/data/object_models/nut_1.obj'],
             'canonical_cloud': pts, 'canonical_normals': nrm,
             'transforms_to_nocs': {'/data/object_models/nut_0.obj': np.eye(4), '/data/object_models/nut_1.obj': np.diag([1.1, 1.1, 0.9, 1.0])},
             'canonical_grasps': np.array(grasps),          # make_canonical.py:126
             'canonical_affordance': rng.uniform(0, 1, len(pts))}
with gzip.open(os.path.join(HERE, 'canonical_golden.pkl'), 'wb') as f:
    pickle.dump(canonical, f)
with gzip.open(os.path.join(HERE, 'complete_grasp_golden.pkl'), 'wb') as f:
    pickle.dump(grasps[:4], f)
np.savez(os.path.join(HERE, 'canonical_golden_expect.npz'), poses=np.stack([g.grasp_pose for g in grasps]),
         scores=np.array([g.perturbation_score for g in grasps]), c1=np.stack([g.c1.point_ for g in grasps]),
         cloud=pts, normals=nrm, affordance=canonical['canonical_affordance'])
print('written', [os.path.getsize(os.path.join(HERE, n)) for n in ('canonical_golden.pkl', 'complete_grasp_golden.pkl')])
