"""Golden vectors for the `use_kdtree_for_eval=True` branch of aligning.estimate9DTransform_worker (aligning.py:63-76).  Build
container only (/root/reference).  The REAL worker runs with two substitutions for packages that cannot be installed here:
cv2.estimateAffine3D on 4 correspondences -> the exact affine through them (as in make_golden_host.py), and open3d's
PointCloud.voxel_down_sample -> oracle.aligning_ref.voxel_down_sample (PARITY UNPINNED w.r.t. open3d; the branch only consumes
nearest-neighbour distances to the down-sampled set, so its order is immaterial).  Everything else -- Utils.toOpen3dCloud, to_homo,
scipy's cKDTree queries, the two-sided error vector, ratio, inliers -- is the reference's own code.

    python tests/golden/make_golden_aligning_kd.py      ->  tests/golden/aligning_kd_golden.npz
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
from catgrasp_amd import synth          # noqa: E402
from oracle import aligning_ref as aref  # noqa: E402


class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ('cv2', 'torchvision', 'trimesh', 'autolab_core', 'pybullet', 'mayavi', 'pybullet_tools', 'pyrender', 'imgaug', 'skimage', 'ikfast_pybind',
             'my_cpp', 'pybullet_env', 'data_reader', 'renderer', 'cvxopt', 'IPython', 'colorlog', 'meshrender', 'perception', 'visualization',
             'shapely', 'networkx', 'pyhull', 'tvtk', 'OpenGL', 'pyglet', 'rtree', 'transformations')

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


class _PointCloud:
    """Functional stand-in for o3d.geometry.PointCloud: the two members the branch touches."""

    def __init__(self):
        self.points = np.zeros((0, 3))

    def voxel_down_sample(self, voxel_size):
        out = _PointCloud()
        out.points = aref.voxel_down_sample(np.asarray(self.points), voxel_size)
        return out


o3d = types.ModuleType('open3d')
o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud)
o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, dtype=np.float64))
sys.modules['open3d'] = o3d
sys.meta_path.insert(0, StubFinder())
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/meshpy')
import aligning  # noqa: E402


def affine4(source, target, confidence=None, ransacThreshold=None):
    Mx = np.concatenate([source, np.ones((4, 1))], axis=1)
    X = np.linalg.solve(Mx, target)
    return 1, np.concatenate([X[:3].T, X[3].reshape(3, 1)], axis=1), np.ones((4, 1))


aligning.cv2.estimateAffine3D = affine4
rng = np.random.default_rng(11)
n = 700
nocs = rng.uniform(-0.5, 0.5, (n, 3)); R = synth.random_rotation(rng); s = np.array([0.016, 0.02, 0.007]); t = np.array([0.02, -0.03, 0.62])
obs = nocs @ (R @ np.diag(s)).T + t + rng.normal(0, 2e-4, (n, 3))
bad = rng.random(n) < 0.25; nocs[bad] = rng.uniform(-0.5, 0.5, (bad.sum(), 3))
ids = np.stack([rng.choice(n, 4, replace=False) for _ in range(160)])
out = {'src': nocs, 'dst': obs, 'ids': ids, 'settings': np.array([[0.003, 0.003], [0.0008, 0.0015]])}      # (PassThreshold, kdtree_eval_resolution)
for si, (thr, res) in enumerate(out['settings']):
    ratios, tfs, inl = [], [], []
    for k in range(len(ids)):
        r = aligning.estimate9DTransform_worker(nocs[ids[k]], obs[ids[k]], nocs, obs, thr, use_kdtree_for_eval=True, kdtree_eval_resolution=res,
                                                max_scale=np.array([0.05] * 3), min_scale=np.array([0.005, 0.005, 0.001]), max_dimensions=np.array([1.2] * 3))
        ratios.append(-1.0 if r[0] is None else r[0]); tfs.append(np.zeros((4, 4)) if r[0] is None else r[1])
        m = np.zeros(n, dtype=np.uint8)
        if r[0] is not None:
            m[r[2]] = 1
        inl.append(m)
    assert (np.array(ratios) >= 0).sum() >= 20
    out[f'ratio{si}'] = np.array(ratios); out[f'tf{si}'] = np.array(tfs); out[f'inliers{si}'] = np.array(inl)
    print('setting', thr, res, ': accepted', int((np.array(ratios) >= 0).sum()), 'of', len(ids), '; ratios', np.round(np.sort(np.array(ratios)[np.array(ratios) >= 0])[-5:], 4))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'aligning_kd_golden.npz'), **out)
