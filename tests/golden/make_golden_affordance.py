"""Generate tests/golden/affordance_golden.npz with the REAL reference `get_finger_contact_area`
(pybullet_env/env_grasp.py:243-283).  Build container only.  Uninstallable imports are inert stubs; open3d is replaced by
a 15-line functional stand-in (PointCloud with points / normals and `transform`, which rotates normals by the upper 3x3
block exactly like open3d's TransformNormals) because the function routes its points through it."""
import importlib
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
    roots = ('cv2', 'torchvision', 'trimesh', 'autolab_core', 'pybullet', 'pybullet_data', 'mayavi', 'pybullet_tools', 'pyrender', 'imgaug',
             'skimage', 'ikfast_pybind', 'my_cpp', 'data_reader', 'renderer', 'camera', 'utils_pybullet', 'env_base')

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
    def __init__(self):
        self.points = np.zeros((0, 3)); self.normals = np.zeros((0, 3)); self.colors = np.zeros((0, 3))

    def transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.points = np.asarray(self.points) @ T[:3, :3].T + T[:3, 3]
        if len(self.normals):
            self.normals = np.asarray(self.normals) @ T[:3, :3].T
        return self


o3d = types.ModuleType('open3d')
o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud)
o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, dtype=np.float64))
sys.modules['open3d'] = o3d
tf_mod = types.ModuleType('transformations'); tf_mod.__all__ = []
sys.modules['transformations'] = tf_mod
sys.meta_path.insert(0, StubFinder())
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/pybullet_env')
env_grasp = importlib.import_module('env_grasp')

from catgrasp_amd import synth  # noqa: E402

rng = np.random.default_rng(3)
ob = synth.make_scene(1, 2500, 9)[0]
g = synth.make_gripper()
fingers = [g['vertices'][8:16].astype(np.float64), g['vertices'][16:24].astype(np.float64)]
grip_dirs = [[0, -1, 0], [0, 1, 0]]
poses = synth.make_candidates(ob, 60, rng)
fmg = g['gripper_in_grasp']
counts, pts_out = [], []
for P in poses:
    cam_in_finger = np.linalg.inv(fmg) @ np.linalg.inv(P)                       # run_grasp_simulation.py:52
    for i in range(2):
        fm = types.SimpleNamespace(vertices=fingers[i])
        sp, dist = env_grasp.get_finger_contact_area(fm, ob_in_finger=cam_in_finger, ob_pts=ob['xyz'], ob_normals=ob['normal'],
                                                     grip_dir=grip_dirs[i], surface_tol=0.005)
        counts.append(-1 if sp is None else len(sp))
        pts_out.append(np.zeros(3) if sp is None else sp.mean(axis=0))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'affordance_golden.npz')
np.savez_compressed(path, xyz=ob['xyz'], normal=ob['normal'], poses=poses, finger0=fingers[0], finger1=fingers[1],
                    finger_mesh_in_grasp=fmg, counts=np.array(counts).reshape(-1, 2), centroid=np.array(pts_out).reshape(-1, 2, 3))
print('wrote', path, os.path.getsize(path), 'bytes;', (np.array(counts) > 0).sum(), 'finger contacts of', len(counts))
