"""Device form of the reference's PointConeGraspSampler candidate generation
(dexnet/grasping/grasp_sampler.py:155-298): per sampled surface point a local frame from the neighbours' normals, then the
fan-out over approach directions x in-plane rotations x approach depths; optionally centred between the fingers.
Returns poses only; filtering is `catgrasp_amd.my_cpp.filterGraspPose`."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int
_c_long = ctypes.c_long


def cone_grasp_poses(points_for_sample, normals_for_sample, sample_ids, sphere_pts, r_ball, hand_depth, init_bite, approach_step,
                     inplane_step_deg=30.0, center_ob_between_gripper=False, device=None, return_tensor=False):
    """-> (K*(1+S*n_rot)*n_depth, 4, 4) float64 grasp poses in the order of the reference's nested loops
    (sample point, rotation, depth).  `r_ball` is the starting ball radius (3 x cloud resolution in the reference, :160);
    the reference's persistent doubling of r_ball when a point has no usable neighbour (:243-247) is reproduced."""
    if device is None:
        if not torch.cuda.is_available():
            raise L.CatgraspAmdError('catgrasp_amd.grasp_sampler needs a HIP device (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())
    pts = torch.from_numpy(np.ascontiguousarray(np.asarray(points_for_sample, dtype=np.float64).reshape(-1, 3))).to(device)
    nrm = torch.from_numpy(np.ascontiguousarray(np.asarray(normals_for_sample, dtype=np.float64).reshape(-1, 3))).to(device)
    ids = torch.from_numpy(np.ascontiguousarray(np.asarray(sample_ids, dtype=np.int32).reshape(-1))).to(device)
    sph = torch.from_numpy(np.ascontiguousarray(np.asarray(sphere_pts, dtype=np.float64).reshape(-1, 3))).to(device)
    P, K, S = pts.shape[0], ids.shape[0], sph.shape[0]
    n_rot = len(np.arange(0, 180, inplane_step_deg))
    n_depth = len(np.arange(0, hand_depth, approach_step))
    total = K * (1 + S * n_rot) * n_depth
    out = torch.empty((total, 16), dtype=torch.float64, device=device)
    if total == 0:
        return out.view(0, 4, 4) if return_tensor else out.view(0, 4, 4).cpu().numpy()
    lib = L.lib()
    dbl = torch.zeros((K,), dtype=torch.int32, device=device)
    check(lib.cg_cone_frames(_p(pts), _p(nrm), _c_int(P), _p(ids), _c_int(K), None, ctypes.c_double(float(r_ball)), _c_int(0), _p(dbl), None,
                             _stream()), 'cg_cone_frames')
    # the reference mutates self.params['r_ball'] while it walks the points in order: radius of point k = r0 * 2^(max doublings so far)
    radii = float(r_ball) * np.power(2.0, np.maximum.accumulate(dbl.cpu().numpy().astype(np.float64)))
    d_r = torch.from_numpy(radii).to(device)
    frames = torch.empty((K, 9), dtype=torch.float64, device=device)
    check(lib.cg_cone_frames(_p(pts), _p(nrm), _c_int(P), _p(ids), _c_int(K), _p(d_r), ctypes.c_double(float(r_ball)), _c_int(1), None, _p(frames),
                             _stream()), 'cg_cone_frames')
    check(lib.cg_cone_poses(_p(pts), _p(ids), _p(frames), _c_int(K), _p(sph), _c_int(S), _c_int(n_rot), ctypes.c_double(float(inplane_step_deg)),
                            _c_int(n_depth), ctypes.c_double(float(approach_step)), ctypes.c_double(float(init_bite)), _p(out), _stream()),
          'cg_cone_poses')
    if center_ob_between_gripper:
        check(lib.cg_center_grasps(_p(out), _c_long(total), _p(pts), _c_int(P), _stream()), 'cg_center_grasps')
    out = out.view(total, 4, 4)
    return out if return_tensor else out.cpu().numpy()
