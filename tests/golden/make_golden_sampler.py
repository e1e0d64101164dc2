"""Golden vectors for the class-level candidate generator: the REAL reference `PointConeGraspSampler.sample_grasps`
(dexnet/grasping/grasp_sampler.py:155-222, incl. Utils.compute_cloud_resolution :492-501 and Utils.hinter_sampling :293-360) run
under a fixed numpy seed with `my_cpp.filterGraspPose` replaced by the identity, i.e. the complete pre-filter candidate list.
Build container only (imports /root/reference under the inert stubs of make_golden_host.py).

    python tests/golden/make_golden_sampler.py   ->   tests/golden/sampler_golden.npz
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ('cv2', 'torchvision', 'open3d', 'trimesh', 'autolab_core', 'pybullet', 'mayavi', 'pybullet_tools', 'pyrender', 'imgaug',
             'skimage', 'ikfast_pybind', 'my_cpp', 'pybullet_env', 'data_reader', 'renderer', 'cvxopt', 'IPython', 'colorlog', 'meshrender',
             'perception', 'visualization', 'shapely', 'networkx', 'pyhull', 'tvtk', 'OpenGL', 'pyglet', 'rtree')

    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name; m.__path__ = []; m.__spec__ = spec; m.__all__ = []
        return m

    def exec_module(self, module):
        pass


def euler_matrix(ai, aj, ak, axes='sxyz'):
    """transformations.euler_matrix, static xyz: R = Rz(ak) Ry(aj) Rx(ai)."""
    assert axes == 'sxyz'
    cx, sx, cy, sy, cz, sz = np.cos(ai), np.sin(ai), np.cos(aj), np.sin(aj), np.cos(ak), np.sin(ak)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    M = np.eye(4); M[:3, :3] = Rz @ Ry @ Rx
    return M


tf_mod = types.ModuleType('transformations')
tf_mod.euler_matrix = euler_matrix
tf_mod.__all__ = ['euler_matrix']
sys.modules['transformations'] = tf_mod
sys.meta_path.insert(0, StubFinder())
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/meshpy')

from dexnet.grasping import grasp_sampler as ref_sampler  # noqa: E402

from catgrasp_amd import synth  # noqa: E402

ref_sampler.my_cpp.filterGraspPose = lambda grasp_poses, *a, **k: grasp_poses        # identity: keep every candidate
rng = np.random.default_rng(77)
pts, nrm = synth.nut_surface(6000, rng)          # dense enough that a 3 x resolution ball holds a few dozen normals
T = np.eye(4); T[:3, :3] = synth.random_rotation(rng); T[:3, 3] = [0.02, -0.01, 0.6]
pts = pts @ T[:3, :3].T + T[:3, 3]; nrm = nrm @ T[:3, :3].T
nrm = nrm + rng.normal(0, 0.05, nrm.shape)          # estimated normals are noisy: on exactly flat faces the scatter matrix is rank 1
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)    # and the minor axis arbitrary
gripper = types.SimpleNamespace(hand_depth=0.04, init_bite=0.005, get_grasp_pose_in_gripper_base=lambda: np.eye(4),
                                trimesh=types.SimpleNamespace(vertices=np.zeros((3, 3)), faces=np.zeros((1, 3), int)),
                                trimesh_enclosed=types.SimpleNamespace(vertices=np.zeros((3, 3)), faces=np.zeros((1, 3), int)))
config = {'sampling_friction_coef': 0.5, 'num_cone_faces': 8, 'grasp_samples_per_surface_point': 1, 'target_num_grasps': 10,
          'min_num_grasps': 10, 'min_contact_dist': 0.0}
out = {'pts': pts, 'nrm': nrm}
for tag, center in (('plain', False), ('centred', True)):
    s = ref_sampler.PointConeGraspSampler(gripper, config)
    np.random.seed(4242)
    grasps = s.sample_grasps(background_pts=np.zeros((1, 3)), points_for_sample=pts.copy(), normals_for_sample=nrm.copy(), max_num_samples=8,
                             n_sphere_dir=5, approach_step=0.01, ee_in_grasp=np.eye(4), cam_in_world=np.eye(4), upper=[0] * 7, lower=[0] * 7,
                             open_gripper_collision_pts=pts, center_ob_between_gripper=center, filter_ik=False, adjust_collision_pose=False)
    out[f'post_call_draws_{tag}'] = np.random.randint(0, 2 ** 31, 4)     # numpy's GLOBAL generator as the reference leaves it (:183, :226)
    out[f'poses_{tag}'] = np.stack([g.grasp_pose for g in grasps])
    out[f'r_ball_{tag}'] = s.params['r_ball']
# NocsTransferGraspSampler.__init__ (grasp_sampler.py:302-327): score threshold, best-n, y-centring -- with the REAL grasp class
rg = np.random.default_rng(8)
can = []
for i in range(12):
    Tg = np.eye(4); Tg[:3, :3] = synth.random_rotation(rg); Tg[:3, 3] = rg.normal(0, 0.01, 3)
    can.append(ref_sampler.ParallelJawPtGrasp3D(grasp_pose=Tg, perturbation_score=float(rg.uniform())))
out['transfer_in_poses'] = np.stack([g.grasp_pose for g in can]); out['transfer_in_scores'] = np.array([g.perturbation_score for g in can])
ts = ref_sampler.NocsTransferGraspSampler(gripper, config, {'canonical_grasps': can}, 'nut', score_larger_than=0.3, max_n_grasp=5,
                                          center_ob_between_gripper=True)
out['transfer_kept_poses'] = np.stack([g.get_grasp_pose_matrix() for g in ts.canonical['canonical_grasps']])
out['transfer_kept_scores'] = np.array([g.perturbation_score for g in ts.canonical['canonical_grasps']])
for cls in ('nut', 'hnm', 'screw'):
    out[f'symmetry_{cls}'] = np.stack(ref_sampler.get_symmetry_tfs(cls))          # Utils.get_symmetry_tfs (Utils.py:79-94) via `from Utils import *`
np.random.seed(99)
out['resolution_seed99'] = ref_sampler.compute_cloud_resolution(pts)
out['hinter_1000'] = ref_sampler.hinter_sampling(min_n_pts=1000, radius=1)[0]
path = os.path.join(ROOT, 'tests', 'golden', 'sampler_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes;', {k: np.shape(v) for k, v in out.items()})
