"""Host-side mirror of the `RobotGripper` fields the scoring / collision path consumes
(dexnet/grasping/gripper.py:55-131), loaded from the reference's gripper directory layout without trimesh /
autolab_core:

    gripper_air_tight.obj, gripper_enclosed_air_tight.obj, finger1.obj [, finger2.obj]     triangle meshes
    params.json                                                                              hand_depth, init_bite, ...
    T_grasp_gripper.tf                                                                       autolab_core RigidTransform text
    gripper_air_tight.sdf, gripper_enclosed_air_tight.sdf                                    optional meshpy SDF text files

Attributes mirror the reference (`trimesh.vertices/.faces`, `trimesh_enclosed`, `get_grasp_pose_in_gripper_base()`,
`finger_{x,y,z}{min,max}`, `get_points_between_finger`, `sdf`, `sdf_enclosed`, one attribute per params.json key);
`filter_args()` returns the four mesh arguments of `my_cpp.filterGraspPose` (common.h:60) in call order."""
import json
import os
import types

import numpy as np


def load_obj(path):
    """Minimal Wavefront OBJ reader: `v x y z` and `f a b c ...` (1-based, `a/b/c` accepted, polygons fan-triangulated,
    negative indices relative to the end).  -> (V (nv,3) float64, F (nf,3) int32)."""
    V, F = [], []
    with open(path, 'r') as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                V.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == 'f':
                idx = [int(s.split('/')[0]) for s in t[1:]]
                idx = [i - 1 if i > 0 else len(V) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    F.append([idx[0], idx[k], idx[k + 1]])
    V = np.asarray(V, dtype=np.float64).reshape(-1, 3)
    F = np.asarray(F, dtype=np.int32).reshape(-1, 3)
    if len(F) and (F.min() < 0 or F.max() >= len(V)):
        raise ValueError(f'{path}: face index out of range')
    return V, F


def save_obj(path, V, F):
    with open(path, 'w') as f:
        for v in np.asarray(V):
            f.write(f'v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n')
        for t in np.asarray(F):
            f.write(f'f {int(t[0]) + 1} {int(t[1]) + 1} {int(t[2]) + 1}\n')


def load_rigid_transform(path):
    """autolab_core `RigidTransform.save` text format: from_frame / to_frame / translation / three rotation rows.
    -> (from_frame, to_frame, 4x4 float64)."""
    with open(path, 'r') as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    if len(lines) < 6:
        raise ValueError(f'{path}: not a RigidTransform .tf file')
    T = np.eye(4)
    T[:3, 3] = [float(x) for x in lines[2].split()]
    T[:3, :3] = [[float(x) for x in lines[3 + r].split()] for r in range(3)]
    return lines[0], lines[1], T


def save_rigid_transform(path, T, from_frame, to_frame):
    T = np.asarray(T, dtype=np.float64)
    with open(path, 'w') as f:
        f.write(f'{from_frame}\n{to_frame}\n')
        f.write(' '.join(repr(float(x)) for x in T[:3, 3]) + '\n')
        for r in range(3):
            f.write(' '.join(repr(float(x)) for x in T[r, :3]) + '\n')


def _mesh(V, F):
    return types.SimpleNamespace(vertices=V, faces=F)


class RobotGripper:
    def __init__(self, gripper_folder, mesh_filename, params, T_grasp_gripper, sdf=None, sdf_enclosed=None):
        """T_grasp_gripper: 4x4, gripper frame -> grasp canonical frame (gripper.py:55-76)."""
        self.gripper_folder = gripper_folder
        self.mesh_filename = mesh_filename
        self.trimesh = _mesh(*load_obj(mesh_filename))
        self.trimesh_enclosed = _mesh(*load_obj(mesh_filename.replace('_air_tight.obj', '_enclosed_air_tight.obj')))
        self.T_grasp_gripper = np.asarray(T_grasp_gripper, dtype=np.float64)
        self.sdf = sdf
        self.sdf_enclosed = sdf_enclosed
        fv, ff = load_obj(os.path.join(os.path.dirname(mesh_filename), 'finger1.obj'))
        T = np.linalg.inv(self.get_grasp_pose_in_gripper_base())
        fv = fv @ T[:3, :3].T + T[:3, 3]
        self.finger_mesh1 = self.finger_mesh1_in_grasp = _mesh(fv, ff)       # trimesh.apply_transform works in place (:66)
        self.finger_xmin, self.finger_xmax = fv[:, 0].min(), fv[:, 0].max()
        self.finger_zmin, self.finger_zmax = fv[:, 2].min(), fv[:, 2].max()
        self.finger_ymin = fv[:, 1].max()          # NOTE (reference :71-72): y is the closing direction; ymin is finger 1's inner...
        self.finger_ymax = -self.finger_ymin       # ...face as written there, mirrored for ymax -- reproduced verbatim
        for key, value in list(params.items()):
            setattr(self, key, value)

    def get_grasp_pose_in_gripper_base(self):
        """inverse of T_grasp_gripper (gripper.py:78-82)"""
        return np.linalg.inv(self.T_grasp_gripper)

    def get_points_between_finger(self, pts_in_grasp):
        p = np.asarray(pts_in_grasp)
        keep = ((p[:, 0] >= self.finger_xmin) & (p[:, 0] <= self.finger_xmax) & (p[:, 1] >= self.finger_ymin) & (p[:, 1] <= self.finger_ymax)
                & (p[:, 2] >= self.finger_zmin) & (p[:, 2] <= self.finger_zmax))
        return p[keep]

    def filter_args(self):
        """(gripper_vertices, gripper_faces, gripper_enclosed_vertices, gripper_enclosed_faces) as filterGraspPose takes them
        (grasp_sampler.py:216-219)."""
        return (self.trimesh.vertices, self.trimesh.faces, self.trimesh_enclosed.vertices, self.trimesh_enclosed.faces)

    @staticmethod
    def load(gripper_dir, device=None, load_sdf=True):
        """gripper.py:92-131.  `gripper_dir` is used as given (the reference resolves it relative to its own source tree).
        SDF grids are uploaded to `device` (catgrasp_amd.sdf.Sdf3D) when the .sdf files exist and a HIP device is present."""
        mesh_filename = os.path.join(gripper_dir, 'gripper_air_tight.obj')
        with open(os.path.join(gripper_dir, 'params.json'), 'r') as f:
            params = json.load(f)
        frm, to, T = load_rigid_transform(os.path.join(gripper_dir, 'T_grasp_gripper.tf'))
        if frm == 'gripper' and to == 'grasp':
            pass
        elif frm == 'grasp' and to == 'gripper':
            T = np.linalg.inv(T)
        else:
            raise RuntimeError('T_grasp_gripper from={}, to={}'.format(frm, to))
        g = RobotGripper(gripper_dir, mesh_filename, params, T)
        if load_sdf:
            from .sdf import SdfFile
            for attr, path in (('sdf', mesh_filename.replace('.obj', '.sdf')),
                               ('sdf_enclosed', mesh_filename.replace('_air_tight.obj', '_enclosed_air_tight.sdf'))):
                if os.path.exists(path):
                    setattr(g, attr, SdfFile(path).read(device=device))
        return g
