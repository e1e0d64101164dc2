"""HBM-roofline table of the scan / gather / write kernels around the two networks ("fraction of HBM roofline is
meaningful only for the scan/gather kernels").  For each kernel: ALGORITHMIC bytes per launch (stated below) / the launch
time measured with HIP events on the launch stream, against the 8 TB/s HBM3E peak (MI355X_MICROARCH.md).  Latency-bound
kernels (FPS: sequential dependency over npoint; the collision filter: L2/LDS gathers) are reported with their own unit.

    python scripts/hbm_kernels.py > profiles/r3_hbm_kernels.json          (on the GPU box)
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import my_cpp, ops, pointgroup_ops, primitives, synth, transforms   # noqa: E402

PEAK = 8000.0   # GB/s
dev = torch.device('cuda:0')


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


rows = []


def row(name, secs, nbytes, what, bound='hbm', extra=None):
    r = {'kernel': name, 'bound': bound, 'avg_launch_us': round(secs * 1e6, 2), 'algorithmic_bytes': int(nbytes),
         'achieved_GBps': round(nbytes / secs / 1e9, 1), 'frac_of_8TBps': round(nbytes / secs / 1e9 / PEAK, 4), 'bytes_are': what}
    if extra:
        r.update(extra)
    rows.append(r)


g = torch.Generator(device=dev); g.manual_seed(0)
# --- candidate input build (dataset_grasp.py:63-91 on the device): 20k-pt scene, 10k candidates x 2048 points
M, G, NP = 20000, 10000, 2048
xyz = torch.randn(M, 3, device=dev, generator=g) * 0.05
nrm = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=1)
obj = (torch.arange(G, device=dev) * 8 // G)[:, None]                       # candidates grouped per object, as bench.py / pipeline.py issue them
ids = (obj * 2500 + torch.randint(0, 2500, (G, NP), device=dev, generator=g)).int().contiguous()      # each candidate resamples its object's cloud
pinv = torch.randn(G, 12, device=dev, generator=g)
out = torch.empty(G, NP, 6, device=dev)
row('build_grasp_input_kernel', timed(lambda: ops.build_grasp_input(xyz, nrm, ids, pinv, out=out)),
    G * (NP * 4 + 48 + NP * 24) + M * 24, 'ids 4 B/pt + pose 48 B + 24 B/pt written + the 480 KB cloud once (object slices are staged in LDS from L2)')
# --- NUNOCS decode: 8 clouds x 8192 points x 300 logits
P = 8 * 8192
lg = torch.randn(P, 300, device=dev, generator=g)
row('nunocs_decode_kernel', timed(lambda: ops.nunocs_decode(lg, 100)), P * (1200 + 16), '1200 B logits in + 16 B out per point')
# --- softmax / p_G over 10k candidates
l10 = torch.randn(G, 10, device=dev, generator=g)
row('softmax_pg_kernel', timed(lambda: ops.softmax_pg(l10)), G * (40 + 52), '40 B in + 52 B out per candidate (launch-latency bound at this size)')
# --- NUNOCS input build: 8 clouds x 8192
ids8 = torch.randint(0, 2500, (8, 8192), device=dev, generator=g).int() + (torch.arange(8, device=dev) * 2500).int()[:, None]
row('build_nunocs_input_kernel', timed(lambda: ops.build_nunocs_input(xyz, nrm, ids8.contiguous())), 8 * 8192 * (4 + 24),
    '4 B id + 24 B out per point; 8 workgroups (one per cloud: min/max reduction) -> latency bound')
# --- PointNet++ primitives at the SURVEY 8(a) sizes: N = 20000, S = 1024, nsample = 32
N, S, K = 20000, 1024, 32
pts = (torch.rand(1, N, 3, device=dev, generator=g) * 0.1).contiguous()
feat = torch.randn(1, N, 6, device=dev, generator=g)
new = pts[:, :S].contiguous()
row('square_distance_kernel', timed(lambda: primitives.square_distance(new, pts)), S * N * 4 + (S + N) * 12, 'S x N x 4 B written')
idx = torch.randint(0, N, (1, S, K), device=dev, generator=g)
row('index_points (group) kernel', timed(lambda: primitives.index_points(feat, idx)), S * K * (8 + 24), '8 B index + 24 B row written per neighbour (gathers are L2 hits)')
t_fps = timed(lambda: primitives.farthest_point_sample(pts, S, start=torch.zeros(1, dtype=torch.long, device=dev)), iters=5, warm=1)
row('fps_blob_kernel (farthest_point_sample)', t_fps, N * 24 + S * 8, 'N x 12 B read twice (box, cells) + gathered once + S x 8 B out', bound='latency chain of one CU (sequential over npoint)',
    extra={'rounds_per_s': round(S / t_fps), 'us_per_round': round(t_fps / S * 1e6, 3),
           'note': 'one 512-thread workgroup, 40 points per thread in VGPRs in spatial order; a round updates only the 512-point blobs within the sampling radius of the new centre (~21 % here), 5.5 VALU per updated point; reference CPU: 0.24 s'})
pts8 = (torch.rand(8, N, 3, device=dev, generator=g) * 0.1).contiguous()
t_fps8 = timed(lambda: primitives.farthest_point_sample(pts8, S, start=torch.zeros(8, dtype=torch.long, device=dev)), iters=5, warm=1)
row('fps_blob_kernel (the 8 clouds of C3 in one launch)', t_fps8, 8 * (N * 24 + S * 8), 'as above x 8 clouds, one workgroup each', bound='latency chain of one CU per cloud',
    extra={'us_per_round_all_clouds': round(t_fps8 / S * 1e6, 3)})
t_bq = timed(lambda: primitives.query_ball_point(0.02, K, pts, new), iters=10)
row('query_ball_point_kernel', t_bq, (N + S) * 12 + S * K * 8, '(N+S) x 12 B read + S x nsample x 8 B written (HBM level)', bound='L2 scan',
    extra={'cache_level_GBps': round(S * N * 12 / t_bq / 1e9, 1), 'cache_level_bytes': 'S x N x 12 B distance tests'})
# --- voxelisation + collision filter on the bench scene
objs = synth.make_scene(8, 2500, seed=0)
gr = synth.make_gripper()
cloud = np.concatenate([o['xyz'] for o in objs]).astype(np.float32)
cl_d = torch.from_numpy(cloud).to(dev)
row('voxel_keys + sort/unique (voxelize)', timed(lambda: my_cpp.voxelize(cl_d, 0.0005, dev), iters=10), len(cloud) * (12 + 8),
    '12 B point in + 8 B key out', bound='launch latency (20k points)')
bg = synth.background_points(objs, 0, gr['diameter'])
sc = my_cpp.GripperScene(gr['vertices'], gr['faces'], gr['enclosed_vertices'], gr['enclosed_faces'], objs[0]['xyz'], bg, 0.0005, dev)
Pn = 50000
Pc = torch.from_numpy(synth.make_candidates(objs[0], Pn, np.random.default_rng(0), gr['hand_depth'], gr['init_bite']).astype(np.float32).reshape(-1, 16)).to(dev)
sym = torch.eye(4, device=dev).reshape(1, 16)
I4 = np.eye(4, dtype=np.float32)
t_f = timed(lambda: my_cpp.filter_on_device(sc, Pc, sym, I4, I4, I4, I4, gr['gripper_in_grasp'], True, False, False), iters=10)
row('filter_grasp_pose_kernel (broad-phase grid)', t_f, Pn * (64 + 66), '64 B pose in + 66 B code/pose/nudge out per evaluation (HBM level)',
    bound='L2/LDS gather latency', extra={'evaluations_per_s': round(Pn / t_f), 'voxels_open': int(sc.keys_open.shape[0]), 'voxels_background': int(sc.keys_bg.shape[0]),
                                         'cache_level_bytes': 'every evaluation scans the 8-byte keys of both voxel sets (L2-resident): evaluations x (voxels_open + voxels_background) x 8 B',
                                         'cache_level_GBps': round(Pn / t_f * (int(sc.keys_open.shape[0]) + int(sc.keys_bg.shape[0])) * 8 / 1e9, 1)})
# --- round 2: the host-loop replacements and the fused set-abstraction layer
G50 = 50000
ids_out = torch.empty((G50, 2048), dtype=torch.int32, device=dev)
t_d = timed(lambda: transforms.draw_ids_device(2500, 2048, G50, dev, seed=7, out=ids_out), iters=10)
row('draw_ids_bijection_kernel (resampling draw, 50k candidates)', t_d, G50 * 2048 * 4, '2048 x 4 B ids written per candidate',
    bound='integer VALU (a 12-round Feistel bijection + cycle walking per index: ~170 instructions per application, ~1.6 applications per index)',
    extra={'rows_per_s': round(G50 / t_d), 'note': 'round 6: a keyed permutation per candidate, 8 indices per thread, no LDS / no barrier (rounds 4-5: a bitonic sort of 4096 '
                                                   '(key, index) pairs per row in LDS, 3.08 ms for the same 50k rows); the reference draws these on the host: ~75 us per candidate'})
poses50 = torch.from_numpy(synth.make_candidates(objs[0], 2000, np.random.default_rng(1)).astype(np.float32).reshape(-1, 16)).to(dev).repeat(25, 1).contiguous()
t_pi = timed(lambda: transforms.pose_inverse_rows_device(poses50, np.zeros(3)))
row('pose_inverse_rows_kernel', t_pi, G50 * (64 + 48), '64 B pose in + 48 B rows out per candidate (launch-latency bound at this size)')
Vb, Fb = synth.subdivide(gr['vertices'], gr['faces'], 4)
Vd, Fd = torch.from_numpy(Vb).to(dev), torch.from_numpy(Fb).to(dev)
t_mg = timed(lambda: my_cpp.MeshGrid(Vb, Fb, 0.0005, dev, V_dev=Vd, F_dev=Fd), iters=5, warm=1)
row('mesh_grid_count/fill/sort (9216-triangle gripper)', t_mg, len(Fb) * 36, 'wall time of the whole device build incl. its one read-back', bound='launch latency',
    extra={'note': 'the numpy builder it replaces: 0.22 s per mesh'})
sa = primitives.SetAbstractionWeights([(np.random.default_rng(0).normal(0, 0.2, (64, 9)), np.zeros(64), None), (np.random.default_rng(1).normal(0, 0.1, (64, 64)), np.zeros(64), None),
                                       (np.random.default_rng(2).normal(0, 0.1, (128, 64)), np.zeros(128), None)], 9, dev)
idx_sa = primitives.query_ball_point(0.02, K, pts, new)
idx_sa = torch.where(idx_sa >= N, torch.zeros_like(idx_sa), idx_sa)
t_sa = timed(lambda: primitives.group_mlp_max(pts, feat, new, idx_sa, sa, check_indices=False))        # kernel alone: no per-call read-back of the error flag
mac = S * K * (9 * 64 + 64 * 64 + 64 * 128)      # algorithmic: the 9 real input channels (the kernel skips the MFMAs of the zero padding to 16)
row('sa_group_mlp_max_kernel (N=20000, S=1024, K=32, mlp 9->64->64->128)', t_sa, S * K * (8 + 36) + S * (12 + 512),
    '8 B index + 36 B gathered row per neighbour + 12 B centroid in + 512 B out per neighbourhood (the unfused pipeline writes and re-reads S x K x (9 + 64 + 64 + 128) x 4 B = 35 MB)',
    bound='mfma (exact f32)', extra={'TFLOPs': round(2 * mac / t_sa / 1e12, 2), 'frac_of_157.3_TFLOPs': round(2 * mac / t_sa / 1e12 / 157.3, 4),
                                     'note': 'register-resident activations, weights in LDS; 1024 neighbourhoods = one 4-wave workgroup per CU: a neighbourhood is one dependent chain of 202 MFMAs'})
Bb = 16
ptsB = (torch.rand(Bb, N, 3, device=dev, generator=g) * 0.1).contiguous(); featB = torch.randn(Bb, N, 6, device=dev, generator=g)
newB = ptsB[:, :S].contiguous()
idxB = primitives.query_ball_point(0.02, K, ptsB, newB); idxB = torch.where(idxB >= N, torch.zeros_like(idxB), idxB)
t_sb = timed(lambda: primitives.group_mlp_max(ptsB, featB, newB, idxB, sa, check_indices=False))
row('sa_group_mlp_max_kernel (16 clouds)', t_sb, Bb * (S * K * (8 + 36) + S * (12 + 512)), 'as above x 16 clouds', bound='mfma (exact f32)',
    extra={'TFLOPs': round(2 * mac * Bb / t_sb / 1e12, 2), 'frac_of_157.3_TFLOPs': round(2 * mac * Bb / t_sb / 1e12 / 157.3, 4)})
coords = torch.cat([torch.zeros(100000, 1, dtype=torch.long, device=dev), torch.randint(0, 64, (100000, 3), device=dev, generator=g)], 1)
t_vx = timed(lambda: pointgroup_ops.voxelization_idx(coords, 1, 4), iters=5, warm=1)
row('voxelization_idx (100k points; pack + device sort + scans + fill)', t_vx, 100000 * (32 + 4 + 8), 'wall time of the whole op incl. two read-backs', bound='launch latency / sort')
xyz_c = torch.rand(20000, 3, device=dev, generator=g) * 0.3
bi = torch.zeros(20000, dtype=torch.int32, device=dev); bo = torch.tensor([0, 20000], dtype=torch.int32, device=dev)
nb_idx, nb_sl = pointgroup_ops.ballquery_batch_p(xyz_c, bi, bo, 0.01, 50)
lab = torch.randint(0, 2, (20000,), device=dev, generator=g).int()
t_cc = timed(lambda: pointgroup_ops.bfs_cluster(lab, nb_idx, nb_sl, 10), iters=5, warm=1)
row('bfs_cluster (20k points, label propagation to the fixed point)', t_cc, nb_idx.numel() * 4 + 20000 * 16, 'wall time of the whole op incl. its convergence read-backs', bound='latency (iterative)')
print(json.dumps({'device': torch.cuda.get_device_name(0), 'hbm_peak_GBps': PEAK, 'kernels': rows}, indent=1))
