"""Generate tests/golden/host_golden.npz by running the REAL reference python code for the host-side pieces of the path
(build container only; /root/reference is read-only and absent on the GPU box).

The reference modules import packages that are not installable here (open3d, trimesh, autolab_core, cv2, transformations,
pybullet, ...).  They are only used by code that is NOT exercised below, so they are replaced by inert auto-stubs; the
functions that ARE exercised run the reference's own numpy/scipy/torch code unmodified:
  dataset_grasp.GraspDataset.transform            dataset_grasp.py:63-91
  dataset_nunocs.NunocsIsolatedDataset.transform  dataset_nunocs.py:38-65   (+ augmentations.NormalizeCloud :66-75)
  Utils.to_homo / normalizeRotation / directionVecToRotation               Utils.py:172-178,262-290,396-402
  meshpy Sdf3D._signed_distance / _signed_distance_batch / is_any_points_inside   meshpy/meshpy/sdf.py:312-389
  meshpy SdfFile._read_3d                                                   meshpy/meshpy/sdf_file.py:59-87
  aligning.estimate9DTransform_worker             aligning.py:33-81  (cv2.estimateAffine3D replaced by the exact 4-point
                                                  affine solve -- the one thing OpenCV contributes there)
  dexnet PointConeGraspSampler.sample_one_surface_point   dexnet/grasping/grasp_sampler.py:225-298 (transformations.euler_matrix
                                                  supplied for the 'sxyz' x-rotation it is called with)
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Inert stubs for the packages the reference imports but this container cannot install, and for reference-local
    modules with heavy import-time side effects.  Everything else (numpy, scipy, torch, matplotlib, the reference's own
    files) is the real thing."""
    roots = ('cv2', 'torchvision', 'open3d', 'trimesh', 'autolab_core', 'pybullet', 'mayavi', 'pybullet_tools', 'pyrender', 'imgaug',
             'skimage', 'ikfast_pybind', 'my_cpp', 'pybullet_env', 'data_reader', 'renderer', 'cvxopt', 'IPython', 'colorlog', 'meshrender',
             'perception', 'visualization', 'shapely', 'networkx', 'pyhull', 'tvtk', 'OpenGL', 'pyglet', 'rtree', 'sympy_stub')

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
    """transformations.euler_matrix for the static-xyz convention the reference uses: R = Rz(ak) Ry(aj) Rx(ai)."""
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

import aligning  # noqa: E402
import augmentations  # noqa: E402
import dataset_grasp  # noqa: E402
import dataset_nunocs  # noqa: E402
import Utils  # noqa: E402
from dexnet.grasping import grasp_sampler as ref_sampler  # noqa: E402
from meshpy import sdf as ref_sdf, sdf_file as ref_sdf_file  # noqa: E402

from catgrasp_amd import synth  # noqa: E402
from oracle import sdf_ref  # noqa: E402

out = {}
rng = np.random.default_rng(7)
ob = synth.make_scene(1, 1200, 11)[0]
ob['xyz'][:5, 2] = 0.05                        # points the z >= 0.1 mask removes
poses = synth.make_candidates(ob, 3, rng)
mean = rng.normal(0, 0.002, 6); std = rng.uniform(0.004, 0.3, 6)

# ---- GraspDataset.transform (phase='test') ----
fake = types.SimpleNamespace(cfg={'n_pts': 256, 'mean': mean, 'std': std}, phase='test')
np.random.seed(3)
g_in = []
for p in poses:
    d = dataset_grasp.GraspDataset.transform(fake, {'cloud_xyz': ob['xyz'].copy(), 'cloud_normal': ob['normal'].copy()}, p)
    g_in.append(d['input'])
out['grasp_xyz'] = ob['xyz']; out['grasp_normal'] = ob['normal']; out['grasp_poses'] = poses
out['grasp_mean'] = mean; out['grasp_std'] = std; out['grasp_input'] = np.array(g_in)
fake2 = types.SimpleNamespace(cfg={'n_pts': 2048}, phase='test')            # more points than the cloud: sampling with replacement, no normaliser
np.random.seed(4)
out['grasp_input_replace'] = dataset_grasp.GraspDataset.transform(fake2, {'cloud_xyz': ob['xyz'].copy(), 'cloud_normal': ob['normal'].copy()}, poses[0])['input']

# ---- NunocsIsolatedDataset.transform (phase='test') + NormalizeCloud ----
fake3 = types.SimpleNamespace(cfg={'n_pts': 512}, phase='test')
np.random.seed(5)
d = dataset_nunocs.NunocsIsolatedDataset.transform(fake3, {'cloud_xyz': ob['xyz'].copy(), 'cloud_normal': ob['normal'].copy(),
                                                            'cloud_nocs': np.zeros_like(ob['xyz']), 'cloud_rgb': np.zeros_like(ob['xyz'])})
out['nunocs_input'] = d['input']; out['nunocs_keep_ids'] = d['keep_ids']; out['nunocs_xyz_original'] = d['cloud_xyz_original']
out['normalize_cloud'] = augmentations.NormalizeCloud()({'cloud_xyz': ob['xyz'][:50].copy()})['cloud_xyz']

# ---- Utils helpers ----
dirs = rng.normal(size=(6, 3)); dirs[0] = [1, 0, 0]
out['dir2rot_dirs'] = dirs
out['dir2rot'] = np.array([Utils.directionVecToRotation(direction=v.copy(), ref=np.array([1., 0, 0])) for v in dirs])
M = rng.normal(size=(4, 4)); out['normrot_in'] = M; out['normrot'] = Utils.normalizeRotation(M)
out['to_homo'] = Utils.to_homo(ob['xyz'][:4])

# ---- meshpy Sdf3D lookups on a fake self ----
data, origin, res = sdf_ref.box_sdf_grid([-0.01, -0.004, -0.007], [0.012, 0.006, 0.003], 0.001, 5)
coords = rng.uniform(-3, data.shape[0] + 3, (3, 400)); coords[:, :40] = np.round(coords[:, :40]) + 0.5
fs = types.SimpleNamespace(data_=data, dims_=np.array(data.shape), data_torch=torch.from_numpy(data).float())
out['sdf_data'] = data.astype(np.float32); out['sdf_coords'] = coords
out['sdf_trilinear'] = ref_sdf.Sdf3D._signed_distance(fs, coords.copy())
out['sdf_fast'] = ref_sdf.Sdf3D._signed_distance(fs, coords.copy(), fast=True)
out['sdf_batch'] = ref_sdf.Sdf3D._signed_distance_batch(fs, torch.from_numpy(coords[None].copy()).float()).numpy()
inside = []
for k in range(8):
    c = rng.uniform(-5, data.shape[0] + 5, (3, 30)); inside.append((c, ref_sdf.Sdf3D.is_any_points_inside(fs, c.copy())))
out['sdf_inside_coords'] = np.array([c for c, _ in inside]); out['sdf_inside'] = np.array([bool(r) for _, r in inside])
# SdfFile._read_3d: capture the constructor arguments instead of building a real Sdf3D (needs autolab_core)
path = '/tmp/_golden_box.sdf'
small = data[:5, :4, :3]
with open(path, 'w') as f:
    f.write('5 4 3\n0.1 0.2 0.3\n0.001\n')
    for k in range(3):
        for j in range(4):
            for i in range(5):
                f.write(f'{small[i, j, k]}\n')
captured = {}
with mock.patch.object(ref_sdf_file.sdf, 'Sdf3D', side_effect=lambda d, o, r: captured.update(d=d, o=o, r=r)):
    ref_sdf_file.SdfFile(path)._read_3d()
out['sdffile_text'] = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
out['sdffile_data'] = captured['d']; out['sdffile_origin'] = captured['o']; out['sdffile_res'] = np.array([captured['r']])

# ---- aligning.estimate9DTransform_worker with the exact 4-point affine in place of cv2.estimateAffine3D ----
def affine4(source, target, confidence=None, ransacThreshold=None):
    Mx = np.concatenate([source, np.ones((4, 1))], axis=1)
    X = np.linalg.solve(Mx, target)
    return 1, np.concatenate([X[:3].T, X[3].reshape(3, 1)], axis=1), np.ones((4, 1))
aligning.cv2.estimateAffine3D = affine4
n = 600
nocs = rng.uniform(-0.5, 0.5, (n, 3)); R = synth.random_rotation(rng); s = np.array([0.016, 0.02, 0.007]); t = np.array([0.02, -0.03, 0.62])
obs = nocs @ (R @ np.diag(s)).T + t + rng.normal(0, 1e-4, (n, 3))
bad = rng.random(n) < 0.2; nocs[bad] = rng.uniform(-0.5, 0.5, (bad.sum(), 3))
ids = np.stack([rng.choice(n, 4, replace=False) for _ in range(120)])
ratios, tfs = [], []
for k in range(len(ids)):
    r = aligning.estimate9DTransform_worker(nocs[ids[k]], obs[ids[k]], nocs, obs, 0.003, max_scale=np.array([0.05] * 3),
                                            min_scale=np.array([0.005, 0.005, 0.001]), max_dimensions=np.array([1.2] * 3))
    ratios.append(-1.0 if r[0] is None else r[0]); tfs.append(np.zeros((4, 4)) if r[0] is None else r[1])
out['ransac_src'] = nocs; out['ransac_dst'] = obs; out['ransac_ids'] = ids
out['ransac_ratio'] = np.array(ratios); out['ransac_tf'] = np.array(tfs)

# ---- PointConeGraspSampler.sample_one_surface_point ----
cone_ob = synth.make_scene(1, 900, 5)[0]
pts, nrm = cone_ob['xyz'].copy(), cone_ob['normal'].copy()
local_n = cone_ob['normal'] @ cone_ob['pose'][:3, :3]
side = np.flatnonzero(np.abs(local_n[:, 2]) < 0.1)
sph = rng.normal(size=(4, 3)); sph /= np.linalg.norm(sph, axis=1, keepdims=True)
fs2 = types.SimpleNamespace(params={'r_ball': 0.003, 'debug_vis': False}, approach_step=0.004,
                            gripper=types.SimpleNamespace(hand_depth=0.02, init_bite=0.005))
fs2.sample_one_surface_point = lambda *a, **k: ref_sampler.PointConeGraspSampler.sample_one_surface_point(fs2, *a, **k)
cone_ids = side[[2, 30, 77]]
blocks = []
for sid in cone_ids:
    gs = ref_sampler.PointConeGraspSampler.sample_one_surface_point(fs2, pts[sid].copy(), nrm[sid].copy(), pts, nrm.copy(), None, sph, seed=1)
    blocks.append(np.array([g.grasp_pose for g in gs]))
out['cone_pts'] = pts; out['cone_nrm'] = nrm; out['cone_sphere'] = sph; out['cone_ids'] = cone_ids; out['cone_poses'] = np.array(blocks)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes')
