"""Golden vectors for the iiwa14 closed-form IK (csrc/iiwa_ik.hip; host restatement oracle/iiwa_ik_ref.py) from the REFERENCE's own solver:
the vendored IKFast file compiled where it lies (oracle/build_ref.py -> oracle/_ref/libikfast_ref.so) and driven exactly like
get_ik_within_limits (my_cpp/common.cpp:9-72).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_iiwa_ik.py        ->  tests/golden/iiwa_ik_golden.npz
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref   # noqa: E402

lib = ctypes.CDLL(build_ref.build())
lib.ik_solutions.restype = ctypes.c_int
lib.ik_within_limits.restype = ctypes.c_int


def ref_fk(j):
    t = (ctypes.c_double * 3)(); r = (ctypes.c_double * 9)()
    lib.ik_fk((ctypes.c_double * 7)(*j), t, r)
    T = np.eye(4); T[:3, :3] = np.array(r).reshape(3, 3); T[:3, 3] = np.array(t)
    return T


rng = np.random.default_rng(2024)
UP = np.deg2rad([170, 120, 170, 120, 170, 120, 175]); LO = -UP          # KUKA LBR iiwa14 R820 joint limits
# forward kinematics of the reference on arbitrary joint vectors (all 7 joints free)
fk_joints = rng.uniform(-3.1, 3.1, (64, 7))
fk_poses = np.stack([ref_fk(j) for j in fk_joints])
# poses: reachable (FK of random joints with the free joint at 0), some pushed out of reach, some generic random rigid poses
N = 6000
J = rng.uniform(-3.1, 3.1, (N, 7)); J[:, 2] = 0
poses = np.stack([ref_fk(j) for j in J])
poses[::7, :3, 3] *= 1.6                                                  # mostly unreachable
poses[3::11, :3, 3] += rng.normal(0, 0.05, (len(poses[3::11]), 3))        # perturbed positions, rotation kept
poses32 = poses.astype(np.float32)                                        # what the filter hands over (Eigen::Matrix4f)
within = np.zeros(N, dtype=bool); nsol = np.zeros(N, dtype=np.int32); sols = np.full((N, 8, 7), np.nan)
buf = np.zeros((16, 7))
for i in range(N):
    ptr = np.ascontiguousarray(poses32[i]).ctypes.data_as(ctypes.c_void_p)
    within[i] = bool(lib.ik_within_limits(ptr, (ctypes.c_double * 7)(*UP), (ctypes.c_double * 7)(*LO)))
    n = lib.ik_solutions(ptr, buf.ctypes.data_as(ctypes.c_void_p), 16)
    nsol[i] = n
    sols[i, :min(n, 8)] = buf[:min(n, 8)]
out = os.path.join(ROOT, 'tests', 'golden', 'iiwa_ik_golden.npz')
np.savez_compressed(out, fk_joints=fk_joints, fk_poses=fk_poses, poses32=poses32, upper=UP, lower=LO, within=within, nsol=nsol, sols=sols)
print('wrote', out, os.path.getsize(out), 'bytes; within limits:', int(within.sum()), 'of', N, '; solution counts', np.unique(nsol, return_counts=True))
