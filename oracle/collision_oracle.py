"""TEST INFRASTRUCTURE ONLY (oracle) -- ctypes binding of oracle/libcollision_ref.so (collision_ref.c).
PARITY UNPINNED: see the header of collision_ref.c (FCL/octomap absent, no reference tests)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, 'libcollision_ref.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _DIR, 'libcollision_ref.so'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def voxelize(pts, resolution):
    """-> (n,3) int32 unique sorted (key-32768)."""
    pts = _f(np.asarray(pts).reshape(-1, 3))
    keys = np.zeros((max(len(pts), 1), 3), dtype=np.int32)
    n = lib().cr_voxelize(_fp(pts), ctypes.c_int(len(pts)), ctypes.c_float(resolution), _fp(keys))
    return keys[:n].copy()


def mesh_voxels_collide(V, F, pose, keys, resolution):
    V = _f(V); F = np.ascontiguousarray(F, dtype=np.int32); pose = _f(pose); keys = np.ascontiguousarray(keys, dtype=np.int32)
    return bool(lib().cr_mesh_voxels_collide(_fp(V), ctypes.c_int(len(V)), _fp(F), ctypes.c_int(len(F)), _fp(pose), _fp(keys),
                                             ctypes.c_int(len(keys)), ctypes.c_float(resolution)))


def mesh_mesh_collide(VA, FA, pose_a, VB, FB, pose_b):
    """isAnyCollision for two posed meshes (collision_manager.cpp:93-111): float64 segment-through-triangle tests."""
    VA = _f(VA); VB = _f(VB); FA = np.ascontiguousarray(FA, dtype=np.int32); FB = np.ascontiguousarray(FB, dtype=np.int32)
    pa = _f(np.asarray(pose_a).reshape(16)); pb = _f(np.asarray(pose_b).reshape(16))
    return bool(lib().cr_mesh_mesh_collide(_fp(VA), _fp(FA), ctypes.c_int(len(FA)), _fp(VB), _fp(FB), ctypes.c_int(len(FB)), _fp(pa), _fp(pb)))


def tri_tri_overlap(P, Q):
    P = _f(np.asarray(P).reshape(9)); Q = _f(np.asarray(Q).reshape(9))
    return bool(lib().cr_tri_tri_overlap64(_fp(P), _fp(Q)))


def box_box_overlap(ca, ha, cb, hb, R):
    ca = _f(ca); cb = _f(cb); R = _f(np.asarray(R).reshape(9))
    return bool(lib().cr_box_box_overlap64(_fp(ca), ctypes.c_float(ha), _fp(cb), ctypes.c_float(hb), _fp(R)))


def voxels_voxels_collide(keys_a, res_a, keys_b, res_b, b_in_a):
    """isAnyCollision for two voxelised clouds; b_in_a = inv(pose A) . pose B (4x4)."""
    ka = np.ascontiguousarray(keys_a, dtype=np.int32); kb = np.ascontiguousarray(keys_b, dtype=np.int32)
    rel = _f(np.asarray(b_in_a).reshape(16))
    return bool(lib().cr_voxels_voxels_collide(_fp(ka), ctypes.c_int(len(ka)), ctypes.c_float(res_a), _fp(kb), ctypes.c_int(len(kb)),
                                               ctypes.c_float(res_b), _fp(rel)))


def tri_box_overlap(c, h, a, b, d):
    c, a, b, d = _f(c), _f(a), _f(b), _f(d)
    return bool(lib().cr_tri_box_overlap(_fp(c), ctypes.c_float(h), _fp(a), _fp(b), _fp(d)))


IK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p)


def filter_grasp_pose(grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs, cam_in_world, ee_in_grasp, gripper_in_grasp,
                      filter_approach_dir_face_camera, filter_ik, adjust_collision_pose, gV, gF, eV, eF,
                      gripper_collision_pts, gripper_enclosed_collision_pts, resolution, ik_fn=None):
    """filterGraspPose restatement; returns codes (E,) int8, poses (E,4,4) f32, nudge (E,) int8 in input order."""
    gp = _f(np.asarray(grasp_poses).reshape(-1, 16)); st = _f(np.asarray(symmetry_tfs).reshape(-1, 16))
    E = len(gp) * len(st)
    mats = [_f(np.asarray(m).reshape(16)) for m in (nocs_pose, canonical_to_nocs, cam_in_world, ee_in_grasp, gripper_in_grasp)]
    gV = _f(gV); eV = _f(eV); gF = np.ascontiguousarray(gF, dtype=np.int32); eF = np.ascontiguousarray(eF, dtype=np.int32)
    k1 = voxelize(gripper_collision_pts, resolution); k2 = voxelize(gripper_enclosed_collision_pts, resolution)
    codes = np.zeros((max(E, 1),), dtype=np.int8); nudge = np.zeros((max(E, 1),), dtype=np.int8)
    poses = np.zeros((max(E, 1), 16), dtype=np.float32)
    cb = IK_FN(ik_fn) if ik_fn is not None else ctypes.cast(None, IK_FN)
    lib().cr_filter_grasp_pose(_fp(gp), ctypes.c_int(len(gp)), _fp(st), ctypes.c_int(len(st)), _fp(mats[0]), _fp(mats[1]), _fp(mats[2]),
                               _fp(mats[3]), _fp(mats[4]), ctypes.c_int(int(filter_approach_dir_face_camera)), ctypes.c_int(int(filter_ik)),
                               ctypes.c_int(int(adjust_collision_pose)), cb, None, _fp(gV), ctypes.c_int(len(gV)), _fp(gF),
                               ctypes.c_int(len(gF)), _fp(eV), ctypes.c_int(len(eV)), _fp(eF), ctypes.c_int(len(eF)), _fp(k1),
                               ctypes.c_int(len(k1)), _fp(k2), ctypes.c_int(len(k2)), ctypes.c_float(resolution), _fp(codes), _fp(poses),
                               _fp(nudge))
    return codes[:E], poses[:E].reshape(E, 4, 4), nudge[:E]


def num_threads():
    return lib().cr_num_threads()


def make_occupancy_grid(pts, resolution):
    """makeOccupancyGridFromCloudScan restatement (K is only used by the reference for unused u/v bounds) -> (Q,3) f32
    in lattice order."""
    pts = _f(np.asarray(pts).reshape(-1, 3))
    lib().cr_make_occupancy_grid.restype = ctypes.c_long
    n = lib().cr_make_occupancy_grid(_fp(pts), ctypes.c_int(len(pts)), ctypes.c_float(resolution), None, ctypes.c_long(0))
    out = np.zeros((max(n, 1), 3), dtype=np.float32)
    n2 = lib().cr_make_occupancy_grid(_fp(pts), ctypes.c_int(len(pts)), ctypes.c_float(resolution), _fp(out), ctypes.c_long(n))
    assert n2 == n
    return out[:n]


def cast_ray(pts, resolution, direction, max_range):
    """octomap castRay restatement from the origin over the occupied leaves of pts -> (hit, leaf centre (3,) f32)."""
    pts = _f(np.asarray(pts).reshape(-1, 3)); d = _f(direction); end = np.zeros(3, dtype=np.float32)
    hit = lib().cr_cast_ray(_fp(pts), ctypes.c_int(len(pts)), ctypes.c_float(resolution), _fp(d), ctypes.c_float(max_range), _fp(end))
    return bool(hit), end


def set_occupancy_variant(tie=0, range_last=0, fcoord=0, fdir=0, strict=0):
    lib().cr_set_occupancy_variant(*[ctypes.c_int(int(v)) for v in (tie, range_last, fcoord, fdir, strict)])


_IK_SO = os.path.join(_DIR, '_ref', 'libikfast_ref.so')


def ikfast_available():
    return os.path.exists(_IK_SO)


def ikfast_within_limits(ee_in_base, upper, lower):
    """oracle/_ref: the reference's IKFast solver behind get_ik_within_limits (common.cpp:9-72).
    ee_in_base (E,4,4) float32 -> bool (E,)."""
    l = ctypes.CDLL(_IK_SO)
    ee = _f(np.asarray(ee_in_base).reshape(-1, 16))
    up = np.ascontiguousarray(upper, dtype=np.float64); lo = np.ascontiguousarray(lower, dtype=np.float64)
    return np.array([bool(l.ik_within_limits(_fp(ee[i]), _fp(up), _fp(lo))) for i in range(len(ee))])
