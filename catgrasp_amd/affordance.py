"""Device form of the reference's grasp-affordance step (run_grasp_simulation.py:50-107 `compute_grasp_affordance`,
pybullet_env/env_grasp.py:243-283 `get_finger_contact_area`): P(T|G) for every candidate in one launch instead of a
python loop with a kd-tree query per finger per grasp."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int
_c_long = ctypes.c_long


def _device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise L.CatgraspAmdError('catgrasp_amd.affordance needs a HIP device (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def nearest_neighbor(query, ref, device=None):
    """cKDTree(ref).query(query)[1] (run_grasp_simulation.py:63): (Q,) int32 cuda tensor of nearest ref indices."""
    dev = _device(device)
    q = torch.from_numpy(np.ascontiguousarray(np.asarray(query, dtype=np.float64).reshape(-1, 3))).to(dev)
    r = torch.from_numpy(np.ascontiguousarray(np.asarray(ref, dtype=np.float64).reshape(-1, 3))).to(dev)
    idx = torch.empty((q.shape[0],), dtype=torch.int32, device=dev)
    check(L.lib().cg_nearest_neighbor(_p(q), _c_long(q.shape[0]), _p(r), _c_int(r.shape[0]), _p(idx), _stream()), 'cg_nearest_neighbor')
    return idx


class AffordanceModel:
    """Per-object data of compute_grasp_affordance (run_grasp_simulation.py:78-99), resident on the device:
    canonical cloud (down-sampled) posed into the camera frame, its normals, and each point's looked-up affordance."""

    def __init__(self, canonical_pts_in_cam, canonical_normals_in_cam, canonical_full_pts_in_cam, canonical_affordance, device=None):
        dev = _device(device)
        self.device = dev
        pts = np.ascontiguousarray(np.asarray(canonical_pts_in_cam, dtype=np.float64).reshape(-1, 3))
        nn = nearest_neighbor(pts, canonical_full_pts_in_cam, dev).cpu().numpy()
        aff = np.asarray(canonical_affordance, dtype=np.float64).reshape(-1)[nn]
        self.pts = torch.from_numpy(pts).to(dev)
        self.normals = torch.from_numpy(np.ascontiguousarray(np.asarray(canonical_normals_in_cam, dtype=np.float64).reshape(-1, 3))).to(dev)
        self.aff = torch.from_numpy(np.ascontiguousarray(aff)).to(dev)


def compute_grasp_affordance(model, grasp_poses_in_cam, finger_mesh_in_grasp, finger_vertices, grip_dirs, surface_tol=0.005,
                             return_counts=False):
    """P(T|G) per grasp: (G,) float64 numpy (NaN where the reference would drop the grasp, run_grasp_simulation.py:68-70).
    finger_vertices: list (one per finger) of (nv,3) finger-mesh vertices; grip_dirs: list of (0,+-1,0)."""
    P = np.asarray(grasp_poses_in_cam, dtype=np.float64).reshape(-1, 4, 4)
    G = len(P)
    cif = (np.linalg.inv(np.asarray(finger_mesh_in_grasp, dtype=np.float64))[None] @ np.linalg.inv(P))[:, :3, :].reshape(G, 12) if G else np.zeros((0, 12))
    ext, signs = [], []
    for V, gd in zip(finger_vertices, grip_dirs):
        V = np.asarray(V, dtype=np.float64)
        gd = np.asarray(gd, dtype=np.float64); gd = gd / np.linalg.norm(gd)
        if np.allclose(gd, [0, 1, 0]):
            signs.append(1)
        elif np.allclose(gd, [0, -1, 0]):
            signs.append(-1)
        else:
            raise RuntimeError(f'grip_dir={gd}')                          # env_grasp.py:262
        ext += [V[:, 0].min(), V[:, 0].max(), V[:, 2].min(), V[:, 2].max()]
    nf = len(signs)
    dev = model.device
    d_cif = torch.from_numpy(np.ascontiguousarray(cif)).to(dev)
    out = torch.empty((G,), dtype=torch.float64, device=dev)
    counts = torch.zeros((G, nf), dtype=torch.int32, device=dev) if return_counts else None
    E = (ctypes.c_double * (4 * nf))(*[float(v) for v in ext])
    S = (ctypes.c_int * nf)(*signs)
    check(L.lib().cg_grasp_affordance(_p(d_cif), _c_long(G), _p(model.pts), _p(model.normals), _p(model.aff), _c_int(model.pts.shape[0]),
                                      _c_int(nf), E, S, ctypes.c_double(float(surface_tol)), _p(out), _p(counts), _stream()),
          'cg_grasp_affordance')
    res = out.cpu().numpy()
    return (res, counts.cpu().numpy()) if return_counts else res
