"""Golden vectors for my_cpp.augmentGraspPoses / directionVecToRotation from the REFERENCE'S OWN C++ (my_cpp/common.cpp:75-153)
compiled by oracle/build_ref.py:build_augment into oracle/_ref/libaugment_ref.so (Eigen from the reference tree).
Build container only.

    python tests/golden/make_golden_augment.py   ->   tests/golden/augment_golden.npz

Only the poses generated from the S valid rows of sphere_pts are kept: the reference's loop runs to sphere_pts.size() = 3*S and
reads beyond the matrix for the rest (heap garbage, not reproducible)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

lib = ctypes.CDLL(build_ref.build_augment())
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rng = np.random.default_rng(2024)
out = {}
# directionVecToRotation: random directions + the degenerate ones (parallel / anti-parallel to ref, tiny cross product)
dirs = np.concatenate([rng.normal(size=(200, 3)), [[1, 0, 0], [-1, 0, 0], [3, 0, 0], [1, 1e-7, 0], [-1, 1e-7, 0], [0, 0, 2], [0, -5, 0]]]).astype(np.float32)
refv = np.array([1, 0, 0], dtype=np.float32)
Rs = np.zeros((len(dirs), 9), dtype=np.float32)
for i, d in enumerate(dirs):
    lib.ref_direction_vec_to_rotation(fp(np.ascontiguousarray(d)), fp(refv), fp(Rs[i]))
out['dvr_dirs'] = dirs; out['dvr_R'] = Rs.reshape(-1, 3, 3)
cases = []
for k, (S, rot_step, depth, step, bite) in enumerate([(5, 30.0, 0.04, 0.002, 0.005), (3, 45.0, 0.03, 0.01, 0.0), (8, 60.0, 0.05, 0.003, 0.01),
                                                      (0, 30.0, 0.04, 0.002, 0.005)]):
    q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
    R0 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float32)
    p = rng.normal(0, 0.3, 3).astype(np.float32)
    sph = rng.normal(size=(S, 3)).astype(np.float32)
    sph = sph / np.linalg.norm(sph, axis=1, keepdims=True) if S else sph
    n_rot = int(np.sum(np.arange(0, 1000) * 0 == 0)) and 0
    x_rot, n_rot = np.float32(0), 0
    while x_rot < 180:
        n_rot += 1; x_rot = np.float32(x_rot + np.float32(rot_step))
    d, n_depth = np.float32(0), 0
    while d < np.float32(depth):
        n_depth += 1; d = np.float32(d + np.float32(step))
    n_valid = (1 + S * n_rot) * n_depth
    cap = (1 + 3 * S * n_rot) * n_depth + 16
    buf = np.zeros((cap, 16), dtype=np.float32)
    n = lib.ref_augment_grasp_poses(fp(np.ascontiguousarray(R0)), fp(p), fp(np.ascontiguousarray(sph)), S, ctypes.c_float(rot_step), ctypes.c_float(depth),
                                    ctypes.c_float(step), ctypes.c_float(bite), fp(buf), cap)
    assert n == (1 + 3 * S * n_rot) * n_depth, (n, S, n_rot, n_depth)       # the reference really iterates 3*S "rows"
    out[f'aug{k}_R0'] = R0; out[f'aug{k}_p'] = p; out[f'aug{k}_sphere'] = sph
    out[f'aug{k}_params'] = np.array([rot_step, depth, step, bite], dtype=np.float32)
    out[f'aug{k}_poses'] = buf[:n_valid].reshape(-1, 4, 4).copy()
    out[f'aug{k}_n_reference'] = n
path = os.path.join(ROOT, 'tests', 'golden', 'augment_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes;', {k: np.shape(v) for k, v in out.items()})
