"""Golden vectors for the two HOST-side PointGroup ops of the inference path from the REFERENCE'S OWN C++: the rule-book builder of
pointgroup_ops.voxelization_idx (voxelize.cpp:34-152, called at predicter.py:285) and the queue BFS of pointgroup_ops.bfs_cluster
(bfs_cluster.cpp:33-91, called at pointgroup.py:240,245), compiled by oracle/build_ref.py:build_pointgroup_host into
oracle/_ref/libpointgroup_host_ref.so.  Build container only.

    python tests/golden/make_golden_pointgroup.py   ->   tests/golden/pointgroup_golden.npz
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import pointgroup_ops_ref as ref  # noqa: E402

lib = ctypes.CDLL(build_ref.build_pointgroup_host())
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rng = np.random.default_rng(31)
out = {}
# ---- voxelization_idx: 3 batches, many duplicate coordinates, batches contiguous (like the real loader), random order inside
n = 4000
coords = np.concatenate([rng.integers(0, 3, (n, 1)), rng.integers(0, 11, (n, 3))], axis=1).astype(np.int64)
coords = np.ascontiguousarray(coords[np.argsort(coords[:, 0], kind='stable')])
out['vox_coords'] = coords
for mode in (4, 3, 1, 2):
    input_map = np.zeros((n,), dtype=np.int32); ma = ctypes.c_int(0)
    m = lib.ref_voxelize_idx(fp(coords), n, 4, 3, mode, fp(input_map), ctypes.byref(ma), None, None)
    output_map = np.zeros((m, ma.value + 1), dtype=np.int32); output_coords = np.zeros((m, 4), dtype=np.int64)
    input_map2 = np.zeros((n,), dtype=np.int32)
    m2 = lib.ref_voxelize_idx(fp(coords), n, 4, 3, mode, fp(input_map2), ctypes.byref(ma), fp(output_map), fp(output_coords))
    assert m2 == m and np.array_equal(input_map, input_map2)
    out[f'vox_mode{mode}_input_map'] = input_map; out[f'vox_mode{mode}_output_map'] = output_map; out[f'vox_mode{mode}_output_coords'] = output_coords
uniq = np.unique(coords, axis=0); rng.shuffle(uniq); uniq = np.ascontiguousarray(uniq)
input_map = np.zeros((len(uniq),), dtype=np.int32); ma = ctypes.c_int(0)
output_map = np.zeros((len(uniq), 2), dtype=np.int32); output_coords = np.zeros((len(uniq), 4), dtype=np.int64)
lib.ref_voxelize_idx(fp(uniq), len(uniq), 4, 3, 0, fp(input_map), ctypes.byref(ma), fp(output_map), fp(output_coords))
out['vox_unique_coords'] = uniq; out['vox_mode0_input_map'] = input_map; out['vox_mode0_output_map'] = output_map; out['vox_mode0_output_coords'] = output_coords
# ---- bfs_cluster: blobs + a 200-point chain + scattered singletons, two semantic classes interleaved in space
blobs = [rng.normal(c, 0.02, (m, 3)) for c, m in (((0, 0, 0), 300), ((0.3, 0, 0), 250), ((0, 0.3, 0), 80), ((1, 1, 1), 25))]
chain = np.stack([np.linspace(2, 3.0, 200), np.zeros(200), np.zeros(200)], 1)
xyz = np.concatenate(blobs + [chain, rng.uniform(5, 9, (30, 3))]).astype(np.float32)
xyz = np.ascontiguousarray(xyz[rng.permutation(len(xyz))])
label = rng.integers(0, 2, len(xyz)).astype(np.int32)
npts = len(xyz)
idx, start_len, _ = ref.ballquery_batch_p(xyz, np.zeros(npts, dtype=np.int32), np.array([0, npts], dtype=np.int32), 0.03, 300)
idx = np.ascontiguousarray(idx, dtype=np.int32); start_len = np.ascontiguousarray(start_len, dtype=np.int32)
out['bfs_xyz'] = xyz; out['bfs_label'] = label; out['bfs_idx'] = idx; out['bfs_start_len'] = start_len
for thr in (1, 50):
    s = ctypes.c_int(0)
    nc = lib.ref_bfs_cluster(fp(label), fp(idx), fp(start_len), npts, thr, ctypes.byref(s), None, None)
    ci = np.zeros((s.value, 2), dtype=np.int32); co = np.zeros((nc + 1,), dtype=np.int32)
    lib.ref_bfs_cluster(fp(label), fp(idx), fp(start_len), npts, thr, ctypes.byref(s), fp(ci), fp(co))
    out[f'bfs_thr{thr}_cluster_idxs'] = ci; out[f'bfs_thr{thr}_cluster_offsets'] = co
path = os.path.join(ROOT, 'tests', 'golden', 'pointgroup_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes;', {k: np.shape(v) for k, v in out.items()})
