#!/usr/bin/env python
"""bench.py -- grasp candidates scored + collision-checked per second (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic candidates (catgrasp_amd/workload.py: SceneBatch.score_slice):
  NUNOCS net over the scene's objects
  -> filterGraspPose with the reference's two live call shapes: canonical grasps x category symmetries with
     adjust_collision_pose=True (dexnet/grasping/grasp_sampler.py:345) and cone poses with symmetry=[I] (:216)
  -> on the device: inverse of every resulting grasp_in_cam, the per-candidate resampling draw of GraspDataset.transform,
     the input transform, PointNetCls, softmax, p_G -- every candidate is scored AND collision-checked (fixed work per step)
  -> one all_gather of the packed (p_G, code) records when --gpus > 1 (catgrasp_amd/distributed.py).

Workloads (config.workload):
  C3 (default; BASELINE.json configs[2]): nut clutter pile, 20k-pt scene (8 objects x 2500 pts), 12 nut symmetries, ONE batch of
      --candidates (50,000) candidates.  --gpus N is STRONG scaling of that same batch (contiguous slices over the ranks + the one
      all_gather), so the N = 1 line is the single-GPU bench line and value(N) / value(1) is the speed-up on a fixed job; the weak
      figure (every rank scores its own 50k candidates of the same scene: --scaling weak) is measured in the same run and reported
      under `secondary` when N > 1.
  C4 (--workload C4; configs[3]): screw category, 40k-pt scene (16 x 2500), 72 symmetries, --candidates-total (200,000)
      candidates in TOTAL cut into contiguous slices over the ranks.
  C5 (--workload C5; configs[4]): mixed-category bin, 60k-pt scene (24 x 2500: nut / hnm / screw in turn, 12 / 2 / 72 symmetries, one
      GraspPredicter + NunocsPredicter per category with its own weights), --candidates-total (500,000) candidates in TOTAL cut into
      contiguous slices over the ranks (strong scaling), split-bf16 MFMA arithmetic (--precision bf16x3) unless told otherwise.
All inputs are resident in HBM before the timed region; weights are seeded random (the reference ships no checkpoints).
`value` is measured under --precision (default f32: exact-f32 MFMA, the reference's arithmetic); the split-precision modes of
the product (f16x3, f16fp8x2, bf16x3; opt-in via CATGRASP_AMD_PRECISION) are measured in the same run and reported under `secondary`.

`roofline.traffic` is MEASURED for the run at N = 1 whenever rocprofv3 is on the box (default; --no-pmc-traffic turns it off): after
the timed region two bounded child passes of the same workload run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
passes, counters only).  If a pass fails the line falls back to the labelled constant from profiles/ and says so.
`rccl_selftest` (N = 1): after the timed region the step's records are gathered once more through a ONE-rank `nccl` process group with
the collective forced -- the pad / all_gather_into_tensor / trim of the N-rank job on RCCL itself -- and compared with the ungathered ones.

Launch: python bench.py --gpus 1 --steps K --warmup W
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
        ... bench.py --gpus 8 --workload C4 --candidates-total 200000           (C4)
        ... bench.py --gpus 8 --workload C5                                     (C5)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as np      # noqa: E402
import torch            # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_PER_POINT_ENC = 9 + 384 + 4096 + 8192 + 131072      # encoder pass (mid_mode 2): T3, conv1, .T64, conv2, conv3
PEAK_F32_MFMA_TFLOPS = 157.3                            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0                         # MI355X_MICROARCH.md: BF16/F16 MFMA dense peak (~2.5 PF)
PEAK_HBM_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# Algorithmic HBM bytes per candidate of the encoder-pass kernel: 49152 B x + 16384 B feature transform + 4096 B out.
ALG_HBM_BYTES_PER_CANDIDATE = 49152 + 16384 + 4096
# Measured HBM bytes per candidate of the same kernel: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, B = 4096),
# 2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, / 4096 -- profiles/r2_pmc_hbm_pointmlp_f32.csv (f32 kernel
# of this round) and profiles/r2_pmc_hbm_pointmlp_split.csv (split kernels: same figures as round 1's r1_pmc_pointmlp.csv).  CONSTANTS from those profiles, not counters read
# in this run (PMC collection needs its own rocprofv3 pass).
PMC_HBM_BYTES_PER_CANDIDATE = {'bf16x3': (2 * 133765.0 + 16384.0) * 1024 / 4096, 'f16x3': (2 * 133762.0 + 16384.0) * 1024 / 4096,
                               'f32': (2 * 133937.0 + 16384.0) * 1024 / 4096}
PMC_HBM_BYTES_PER_CANDIDATE['f16fp8x2'] = (2 * 134646.0 + 18420.0) * 1024 / 4096         # profiles/r2_pmc_hbm_pointmlp_f16fp8x2.csv
L3_SHARE = 131072.0 / MAC_PER_POINT_ENC                 # share of the pass's MACs in the 128 -> 1024 layer
DTYPE = {'f32': 'f32 (exact-f32 MFMA: every product and accumulation in float32, as the reference)',
         'f16x3': 'f32 in/out/accumulate; wide-layer products as 3x f16 MFMA on hi+lo half pieces (f16x3 split, 22 significant bits)',
         'bf16x3': 'f32 in/out/accumulate; wide-layer products as 3x bf16 MFMA on hi+lo bf16 pieces (bf16x3 split, 16 significant bits)',
         'f16fp8x2': 'f32 in/out/accumulate; 128->1024 layers as 1 f16 MFMA + 2 block-scaled e4m3 MFMAs (half cost each) per product block, '
                     'all other wide layers f16x3'}


def api_block(batch, gp, device, n=50000):
    """Wall-clock of the REFERENCE entry points exactly as run_grasp_simulation.py:176,310 call them (python lists / numpy arrays in,
    python lists out; includes every host-side conversion, upload, download and list build), next to the device-resident number."""
    from catgrasp_amd import engine, my_cpp, synth
    out = {}
    ob = batch.objs[0]
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    rng = np.random.default_rng(5)
    base = synth.make_candidates(ob, 2000, rng, batch.gripper['hand_depth'], batch.gripper['init_bite'])
    poses = list(base[rng.integers(0, len(base), n)])                 # list of (4,4) float64, like grasps[i].get_grasp_pose_matrix()

    def wall(fn, reps):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), r
    pb = []
    for prec in dict.fromkeys([engine.current_precision(), 'f16x3']):
        with engine.precision(prec):
            gp.predict_batch(data, poses[:2000], rng='device')        # warm-up
            t, ret = wall(lambda: gp.predict_batch(data, poses, rng='device'), 3)
        assert len(ret) == n and len(ret[0]) == 3 and ret[0][2].shape == (10,)
        pb.append({'poses': n, 'rng': 'device', 'precision': prec, 'wall_s': round(t, 4), 'candidates_per_s': round(n / t, 1)})
    for prec in dict.fromkeys([engine.current_precision(), 'f16x3']):
        with engine.precision(prec):
            gp.predict_batch(data, poses[:2000], rng='numpy')         # warm-up of this mode too (worker thread, stream replay, swap-chain kernel)
            ts = []
            for _ in range(2):                                        # every call consumes numpy's stream: reseed, time each call on its own
                np.random.seed(0)
                ts.append(wall(lambda: gp.predict_batch(data, poses, rng='numpy'), 1)[0])
            t = float(np.median(ts))
        pb.append({'poses': n, 'rng': 'numpy (the default: the reference\'s global-generator stream, bit-identical draws and generator state; the '
                                      'sequential rejection sampling of the stream replayed in C on one host core one chunk ahead of the device, '
                                      'the permutation swap chains on the device)',
                   'precision': prec, 'wall_s': round(t, 4), 'candidates_per_s': round(n / t, 1)})
    out['predict_batch'] = pb
    # small calls (what the reference issues per object is a few hundred poses, predicter.py:67-94): wall-clock of a whole predict_batch
    # call at 1 / 16 / 256 poses, median of 15 after 3 warm-ups, exact f32, both draw modes
    small = []
    for G in (1, 16, 256):
        for mode in ('device', 'numpy'):
            ts = []
            for i in range(18):
                np.random.seed(i); torch.cuda.synchronize(); t0 = time.perf_counter()
                gp.predict_batch(data, poses[:G], rng=mode)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            ts = sorted(ts[3:])
            small.append({'poses': G, 'rng': mode, 'median_ms': round(ts[len(ts) // 2] * 1e3, 3), 'min_ms': round(ts[0] * 1e3, 3)})
    out['predict_batch_small_calls'] = small
    # filterGraspPose, 20 positional arguments, >= 5k-triangle gripper meshes, nut symmetries, pose nudging on
    g = batch.gripper
    V, F, Ve, Fe = g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces']       # the batch's gripper: 9,216 / 12,288 triangles
    from catgrasp_amd import transforms
    sym = transforms.get_symmetry_tfs('nut')
    n_can = (n + len(sym) - 1) // len(sym)
    can = list(np.linalg.inv(batch.nocs_pose[0]) @ base[rng.integers(0, len(base), n_can)])
    bg = synth.background_points(batch.objs, 0, g['diameter'])
    I4 = np.eye(4)
    args = (can, sym, batch.nocs_pose[0], I4, I4, I4, g['gripper_in_grasp'], True, False, True, [0] * 7, [0] * 7, V, F, Ve, Fe, ob['xyz'], bg,
            0.0005, False)
    my_cpp.clear_scene_cache()
    t_cold, surv = wall(lambda: my_cpp.filterGraspPose(*args), 1)
    t_warm, surv = wall(lambda: my_cpp.filterGraspPose(*args), 3)
    E = n_can * len(sym)
    out['filterGraspPose'] = {'evaluations': E, 'canonical_grasps': n_can, 'symmetries': len(sym), 'gripper_triangles': [int(len(F)), int(len(Fe))],
                              'adjust_collision_pose': True, 'survivors': len(surv),
                              'cold_wall_s': round(t_cold, 4), 'cold_evaluations_per_s': round(E / t_cold, 1),
                              'warm_wall_s': round(t_warm, 4), 'warm_evaluations_per_s': round(E / t_warm, 1),
                              'note': 'cold = first call on these meshes/clouds (device mesh-grid build + voxelisation); warm = the '
                                      'content-keyed scene cache is hit, as on the second per-object call of grasp_sampler.py'}
    return out


GRIPPER_SUBDIVISIONS = 4                                # the step's gripper meshes: 36 / 48 box triangles x 4^4 = 9,216 / 12,288
PEAK_L2_GBS = 34500.0                                   # MI355X_MICROARCH.md: L2 4 MiB per XCD, ~34.5 TB/s aggregate


def step_filter_rects(batch):
    """The rectangles (segment, i0, i1, j0, j1) of the step's filter: every evaluation of the batch, as score_slice(0, n_total) issues them."""
    from catgrasp_amd import workload
    return [(s, *r) for s, a, b_ in workload.intersect(batch.segs, 0, batch.n_total) for r in workload.split_eval_range(s.n_sym, a, b_)]


def filter_roofline_block(batch, device, reps=5, issue=None):
    """roofline_filter: the collision filter of the STEP -- every segment of every object (canonical grasps x symmetries with nudging; cone
    poses) in the one launch sequence the step issues (cg_filter_grasp_pose_multi: pose composition, grid kernel, exhaustive finisher),
    alone on the stream: HIP-event time of the sequence against the chip's vector-instruction issue rate (`issue` = pmc_filter_issue's
    counters of the same sequence), with, beside it, the CACHE-LEVEL bytes the grid kernel itself counts (work_stats: 8 B per voxel key
    read, 8 B per grid cell looked up, 48 B of triangle + 4 B of list entry per (voxel, triangle) pair tested) plus the algorithmic HBM
    bytes (64 B pose in, 66 B out per evaluation).  HBM-wise the kernel is trivial; the voxel keys, cell table and triangles are L2 / L1
    resident."""
    g = batch.gripper
    rects = step_filter_rects(batch)
    key = ('roofline', 0, batch.n_total)
    batch.run_filter_many(key, rects)                       # builds the plan
    plan = batch._plans[key]
    stats = torch.zeros(3, dtype=torch.int64, device=device)
    out = plan.run(g['gripper_in_grasp'], True, keep_rejected_pose=True, work_stats=stats)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run(g['gripper_in_grasp'], True, keep_rejected_pose=True)
    e1.record(); torch.cuda.synchronize()
    w = stats.cpu().numpy()
    E_tot = int(out[0].numel())
    tot_ms = e0.elapsed_time(e1) / reps
    tot_bytes = int(8 * w[0] + 8 * w[1] + 52 * w[2] + 130 * E_tot)
    gbs = tot_bytes / (tot_ms * 1e-3) / 1e9
    cache = {'cache_level_bytes': tot_bytes, 'cache_level_GBps': round(gbs, 1), 'frac_of_l2_peak': round(gbs / PEAK_L2_GBS, 4), 'l2_peak_GBps': PEAK_L2_GBS,
             'voxel_keys_read': int(w[0]), 'grid_cells_looked_up': int(w[1]), 'pairs_tested': int(w[2]),
             'note': 'what the lanes load from L1 / L2 (counted by the kernel itself), not HBM: 8 B per voxel key, 8 B per grid cell, 52 B per pair'}
    out = {'bound': 'valu-issue', 'kernel': 'compose_grasp_pose_multi_kernel + filter_grasp_pose_kernel<true, true> (+ the exhaustive finisher, which finds nothing to do)',
           'achieved': None, 'peak': round(PEAK_VALU_WAVE_INSTS_PER_S / 1e9, 1), 'unit': 'G wave-instructions/s', 'frac': None, 'traffic': None,
           'launch_sequences_per_step': 1, 'segments': len(rects), 'evaluations': E_tot, 'ms': round(tot_ms, 4),
           'evaluations_per_s': round(E_tot / (tot_ms * 1e-3), 1),
           'gripper_triangles': [int(len(g['faces'])), int(len(g['enclosed_faces']))], 'hbm_algorithmic_bytes': 130 * E_tot, 'cache_level': cache,
           'note': 'the whole step\'s filter (all objects, both call shapes) is ONE launch sequence since round 6 (rounds 3-5: 16 calls x 3 launches of '
                   '~6k evaluations on 8 streams).  The kernel is bound by vector-instruction issue (broad-phase arithmetic per voxel, the 13-axis '
                   'separating-axis test per pair), neither by HBM (130 B per evaluation) nor by cache bytes: `achieved` = wavefront-level VALU '
                   'instructions of the grid kernel (SQ_INSTS_VALU, PMC child pass of this run) / the HIP-event time of the sequence; peak = one wave64 '
                   'instruction per CU per clock (256 CUs x 2.4 GHz); wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES'}
    if issue is not None and issue[0] is not None:
        c = issue[0]
        ach = c['valu_wave_insts'] / (tot_ms * 1e-3)
        out.update(achieved=round(ach / 1e9, 1), frac=round(ach / PEAK_VALU_WAVE_INSTS_PER_S, 4), valu_wave_insts=int(c['valu_wave_insts']),
                   valu_wave_insts_per_evaluation=round(c['valu_wave_insts'] / max(E_tot, 1), 1), wait_share=round(c['wait_any'] / c['wave_cycles'], 4),
                   counters_source=issue[1])
    elif issue is not None:
        out['counters_source'] = f'not measured in this run ({issue[1]})'
    return out


def encoder_block(batch, device, clouds=(1, 8)):
    """The PointNet++ set-abstraction encoder north_star names (catgrasp_amd.pointnet2.PointNet2Encoder: 512 / 0.2 / 32 -> 128 / 0.4 / 64 ->
    all) on the step's own 20k-point scene cloud, normalised into the unit ball, xyz + normals: ms per forward (HIP events over 10
    forwards) and the f32 MFMA fraction of the two fused sampling levels from their own event timings.  Random-init weights."""
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    pts = np.concatenate([o['xyz'] for o in batch.objs]); nrm = np.concatenate([o['normal'] for o in batch.objs])
    pts = pts - pts.mean(0); pts = pts / np.linalg.norm(pts, axis=1).max()
    x1 = torch.from_numpy(np.concatenate([pts, nrm], 1).astype(np.float32)).to(device)
    torch.manual_seed(0)
    enc = p2.PointNet2Encoder(channel=6).to(device).eval()
    was = p2.VALIDATE_INPUTS
    p2.VALIDATE_INPUTS = False
    out = {'model': 'PointNet2Encoder(channel=6): SA(512, r 0.2, K 32, [64,64,128]) -> SA(128, r 0.4, K 64, [128,128,256]) -> SA(all, [256,512,1024])',
           'points': int(len(pts)), 'peak_tflops_f32_mfma': PEAK_F32_MFMA_TFLOPS, 'rows': []}

    def timed(fn, iters=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    try:
        with torch.no_grad():
            for B in clouds:
                x = x1[None].repeat(B, 1, 1).contiguous()
                N = x.shape[1]
                start = (torch.arange(B) * 37 % N, torch.arange(B) * 11 % 512)
                ms = timed(lambda: enc(x, start=start))
                xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
                _, l1_xyz = p2.farthest_point_sample(xyz, 512, start[0], return_xyz=True)
                idx1 = p2.query_ball_point(0.2, 32, xyz, l1_xyz)
                W1, W2 = enc.sa1._weights(device), enc.sa2._weights(device)
                l1 = prim.group_mlp_max(xyz, feats, l1_xyz, idx1, W1, channels_last=True)
                _, l2_xyz = p2.farthest_point_sample(l1_xyz, 128, start[1], return_xyz=True)
                idx2 = p2.query_ball_point(0.4, 64, l1_xyz, l2_xyz)
                t1 = timed(lambda: prim.group_mlp_max(xyz, feats, l1_xyz, idx1, W1, check_indices=False, channels_last=True))
                t2 = timed(lambda: prim.group_mlp_max(l1_xyz, l1, l2_xyz, idx2, W2, check_indices=False, channels_last=True))
                l2 = prim.group_mlp_max(l1_xyz, l1, l2_xyz, idx2, W2, check_indices=False, channels_last=True)
                W3 = enc.sa3._weights(device)
                stages = {'fps 20000 -> 512': timed(lambda: p2.farthest_point_sample(xyz, 512, start[0], return_xyz=True)),
                          'ball query r 0.2 K 32': timed(lambda: p2.query_ball_point(0.2, 32, xyz, l1_xyz)),
                          'level 1 fused 9-64-64-128': t1,
                          'fps 512 -> 128': timed(lambda: p2.farthest_point_sample(l1_xyz, 128, start[1], return_xyz=True)),
                          'ball query r 0.4 K 64': timed(lambda: p2.query_ball_point(0.4, 64, l1_xyz, l2_xyz)),
                          'level 2 fused 131-128-128-256': t2,
                          'level 3 group-all 259-256-512-1024 (3 launches)': timed(lambda: prim.group_all_mlp_max(l2_xyz, l2, W3, fused=False))}
                f1 = B * 512 * 32 * 2 * (9 * 64 + 64 * 64 + 64 * 128); f2 = B * 128 * 64 * 2 * (131 * 128 + 128 * 128 + 128 * 256)
                out['rows'].append({'clouds': B, 'ms_per_forward': round(ms, 4), 'clouds_per_s': round(B / ms * 1e3, 1),
                                    'level1_fused_us': round(t1 * 1e3, 2), 'level1_frac_of_peak': round(f1 / (t1 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    'level2_fused_us': round(t2 * 1e3, 2), 'level2_frac_of_peak': round(f2 / (t2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    'stages_us': {k: round(v * 1e3, 2) for k, v in stages.items()}})
    finally:
        p2.VALIDATE_INPUTS = was
    out['note'] = ('a forward at one cloud is dominated by the two farthest-point-sampling chains (one CU per cloud, ~0.9 us per round: 512 + 128 rounds); '
                   '`stages_us`: every stage alone on the stream, as the module issues it; rocprofv3 kernel statistics: profiles/r5_pp_encoder*.csv')
    return out


def pick_cycle_block(batch, gp, npred, device, n_canonical=2000):
    """One pick cycle's per-object work with the package's DEFAULT settings (exact f32, the reference's numpy stream for the resampling
    draw AND for the 2 x 10,000 RANSAC hypothesis draws), as run_grasp_simulation.py:112-183,296-329 issues it: pipeline.evaluate_object
    over the 8 C3 objects, per-stage wall-clock (host clock, device drained after every stage).  Next to it the same cycle with the
    non-reference fast draws (ransac_sampling='fast', rng='device').  Random-init weights cannot recover a pose, so the canonical-grasp
    branch takes the scene's true NUNOCS pose (the RANSAC still runs and is timed)."""
    from catgrasp_amd import engine, pipeline, synth, transforms
    from catgrasp_amd.predicter import DEFAULT_NUNOCS_CFG, NunocsPredicter
    g = dict(batch.gripper)
    g['finger_vertices'] = [g['vertices'][8:16], g['vertices'][16:24]]
    g['grip_dirs'] = [[0, -1, 0], [0, 1, 0]]
    objs = batch.objs
    scene_pts = np.concatenate([o['xyz'] for o in objs])
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    sym = transforms.get_symmetry_tfs('nut')
    rng = np.random.default_rng(11)
    canon_pts, canon_nrm = synth.nut_surface(3000, rng)
    fast = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=npred.model.state_dict(), device=device, ransac_sampling='fast')

    job = []
    for k, ob in enumerate(objs):
        grasps = np.linalg.inv(batch.nocs_pose[k]) @ synth.make_candidates(ob, n_canonical, np.random.default_rng(100 + k), g['hand_depth'], g['init_bite'])
        job.append({'ob_pts': ob['xyz'], 'ob_normals': ob['normal'], 'symmetry_tfs': sym, 'nocs_pose_override': batch.nocs_pose[k],
                    'canonical': {'cloud': canon_pts, 'normals': canon_nrm, 'affordance': np.linspace(0, 1, 3000), 'grasps': grasps}})

    def cycle(nun, rng_mode, overlap):
        per_ob = []
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = pipeline.evaluate_objects(job, scene_pts, K, g, gp, nun, draw_ahead=bool(overlap), overlap=overlap, timings=per_ob, rng=rng_mode)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        timings = {}
        for tm in per_ob:
            for k_, v in tm.items():
                timings[k_] = timings.get(k_, 0.0) + v
        import hashlib
        h = hashlib.sha256()
        for o in outs:
            h.update(np.ascontiguousarray(o['poses']).tobytes()); h.update(np.ascontiguousarray(o['p_T_G']).tobytes())
        h.update(np.random.get_state()[1].tobytes()); h.update(str(np.random.get_state()[2]).encode())
        return wall, sum(o['n_evaluated'] for o in outs), sum(len(o['poses']) for o in outs), timings, h.hexdigest()[:16]
    res = {'objects': len(objs), 'precision': engine.current_precision()}
    for name, nun, rng_mode, ahead in (('default', npred, 'numpy', 'stages'), ('default_draws_ahead_only', npred, 'numpy', 'draws'),
                                       ('default_serial', npred, 'numpy', None), ('fast_draws', fast, 'device', None)):
        cycle(nun, rng_mode, ahead)                         # warm-up: caches, allocator, worker threads
        wall, total, surv, tm, digest = cycle(nun, rng_mode, ahead)
        res[name] = {'settings': {'default': "pipeline.evaluate_objects: ransac_sampling='reference', rng='numpy' (bit-identical to a seeded reference run), "
                                             "object k+1's whole pre-scoring half (occupancy, NUNOCS + RANSAC incl. its 2 x 10,000 hypothesis draws, cone "
                                             "sampler, filter, affordance) on a second thread + stream while the device scores object k (overlap='stages')",
                                  'default_draws_ahead_only': "the same with only the next object's hypothesis draws made ahead (overlap='draws'): the round-5 default",
                                  'default_serial': 'the same, object after object (overlap=None): the round-4 figure',
                                  'fast_draws': "ransac_sampling='fast', rng='device' (same distributions, not numpy's stream)"}[name],
                     'wall_s_per_cycle': round(wall, 4), 'wall_ms_per_object': round(wall / len(objs) * 1e3, 2),
                     'evaluations': total, 'survivors_scored': surv, 'results_and_generator_state_sha256_16': digest,
                     'ms_per_object_by_stage': {k: round(v / len(objs) * 1e3, 3) for k, v in tm.items()}}
    res['draw_ahead_equals_serial'] = (res['default']['results_and_generator_state_sha256_16'] == res['default_serial']['results_and_generator_state_sha256_16']
                                       == res['default_draws_ahead_only']['results_and_generator_state_sha256_16'])
    res['note_stage_times'] = ("with overlap='stages' the pre-scoring stage times of object k+1 are measured on the second thread, beside object k's scoring "
                               "pass: they overlap it and do not add up to wall_ms_per_object")
    res['note'] = ("stages: occupancy (background ray cast), nunocs+ransac (= 'nunocs net + decode' + 'ransac id draw (exposed)' + 'ransac kernels + "
                   "selection'; 'ransac id draw' is the duration of the stream replay itself -- on the stream worker under the network in the serial loop, "
                   "on the draw-ahead thread under the PREVIOUS object's scoring pass in the default), "
                   'candidate generation (cone sampler), filterGraspPose (cone poses + canonical grasps x 12 symmetries, nudging on), affordance, grasp-Q scoring')
    return res


def projected_scaling_block(batch, n_total, ms_full_s, steps=3):
    """What ONE GPU needs for the slice a rank of an N-rank job would get (strong scaling: contiguous slices of the same batch, rank 0's
    and the last rank's, whichever is slower), measured in this run -> the speed-up the N-rank job can reach before its one all_gather.
    A projection from single-GPU timings: no multi-GPU hardware was available to the builder; the driver's SCALE run is the measurement."""
    from catgrasp_amd import distributed as cgd
    out = {'what': 'single-GPU time of the slice one rank of an N-rank strong-scaling job scores (max of the first and the last rank\'s slice), '
                   'this run; speed-up = full-batch step / slice step, before the all_gather of the 8 B records',
           'status': 'projection -- unmeasured on multi-GPU hardware', 'full_batch_ms': round(ms_full_s * 1e3, 3), 'ranks': {}}
    with torch.no_grad():
        for world in (2, 4, 8):
            _, bounds = cgd.shard_bounds(n_total, world)
            worst = 0.0
            for lo, hi in (bounds[0], bounds[-1]):
                batch.score_slice(lo, hi); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    batch.score_slice(lo, hi)
                torch.cuda.synchronize()
                worst = max(worst, (time.perf_counter() - t0) / steps)
            out['ranks'][str(world)] = {'slice_candidates': bounds[0][1] - bounds[0][0], 'slice_ms': round(worst * 1e3, 3),
                                        'projected_speedup': round(ms_full_s / worst, 3)}
    return out


def ctypes_float0():
    import ctypes
    return ctypes.c_float(0.0)


def cpu_baseline(batch, sd_cls, sd_seg, n_score=80, n_coll=4096):
    """BASELINE.md §3 on this box's host cores, bounded to ~30 s: the reference's op sequence (F.conv1d / F.batch_norm / F.linear
    port, oracle/pointnet_ref.py -- the reference package cannot travel to the GPU box) in chunks of 200 (predicter.py:69) fed by
    the restated per-candidate GraspDataset.transform python loop; the C/OpenMP filterGraspPose restatement over all cores with
    its per-call structure build; one NUNOCS forward per object amortised.  3 warm-ups, median of 3, thread scan logged."""
    from oracle import collision_oracle as co
    from oracle import pointnet_ref as oref
    from oracle import transforms_ref as tref
    from catgrasp_amd import synth as _synth
    ob = batch.objs[0]
    # The step's gripper meshes are the box gripper's surfaces subdivided to 9,216 / 12,288 triangles.  The CPU restatement has no BVH
    # (FCL has), so it is timed on the SAME surfaces un-subdivided (36 / 48 triangles: identical verdicts) and with the float32
    # separating-axis narrow phase -- the cheapest form of the predicate, not the oracle's float64 clipping.
    g = _synth.make_gripper()
    co.lib().cr_set_variant(0, ctypes_float0(), 1)           # restored in the `finally` below
    seg_nocs = next(s for s in batch.segs if s.kind == 'nocs' and s.obj == 0)
    seg_cone = next(s for s in batch.segs if s.kind == 'cone' and s.obj == 0)
    P = batch.host_poses(seg_cone)
    psd = oref.prepared_state_dict(sd_cls)
    host = os.cpu_count()
    scan = {}
    xs = np.stack([tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], tref.draw_ids(len(ob['xyz']), 2048))['input'] for i in range(24)])
    xw = torch.from_numpy(xs).float()
    with torch.no_grad():
        for nt in sorted({min(host, t) for t in (8, 16, 32, 64, 128, host)}):
            torch.set_num_threads(nt)
            oref.pointnet_cls_forward_nnops(psd, xw[:8])
            t0 = time.perf_counter(); oref.pointnet_cls_forward_nnops(psd, xw); scan[nt] = round((time.perf_counter() - t0) / len(xw) * 1e3, 2)
    nthreads = min(scan, key=scan.get)
    torch.set_num_threads(nthreads)

    def transform_loop(n):
        return np.stack([tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], tref.draw_ids(len(ob['xyz']), 2048))['input']
                         for i in range(n)])

    def net(x):
        with torch.no_grad():
            return torch.cat([oref.pointnet_cls_forward_nnops(psd, x[s:s + 200])[0] for s in range(0, len(x), 200)])
    for _ in range(3):
        net(xw)
    t_tr, t_net = [], []
    for _ in range(3):
        t0 = time.perf_counter(); x = torch.from_numpy(transform_loop(n_score)).float(); t1 = time.perf_counter()
        tref.predict_batch_post(net(x).numpy()); t2 = time.perf_counter()
        t_tr.append((t1 - t0) / n_score); t_net.append((t2 - t1) / n_score)
    t_transform, t_netonly = float(np.median(t_tr)), float(np.median(t_net))
    # collision: both call shapes, half the sample each, per-call structure build included (as common.cpp:176-182)
    bg = __import__('catgrasp_amd.synth', fromlist=['x']).background_points(batch.objs, 0, g['diameter'])
    I4 = np.eye(4)
    from catgrasp_amd import transforms
    sym = transforms.get_symmetry_tfs(batch.cats[0])
    n_can = max(1, (n_coll // 2) // len(sym))
    can = batch.host_poses(seg_nocs)[:n_can]
    t0 = time.perf_counter()
    co.filter_grasp_pose(can, sym, batch.nocs_pose[0], I4, I4, I4, g['gripper_in_grasp'], 1, 0, 1, g['vertices'], g['faces'],
                         g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
    co.filter_grasp_pose(P[:n_coll // 2], [I4], I4, I4, I4, I4, g['gripper_in_grasp'], 1, 0, 0, g['vertices'], g['faces'],
                         g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
    t_coll = (time.perf_counter() - t0) / (n_can * len(sym) + n_coll // 2)
    # The SAME call on the step's own subdivided meshes (9,216 / 12,288 triangles), a small sample: what the restatement -- which has
    # no bounding-volume hierarchy where FCL has one -- costs on identical inputs.  Reported beside the figure above, not used in `value`
    # (it would credit the GPU with FCL's missing BVH).
    t_coll_same = None
    try:
        V, F = _synth.subdivide(g['vertices'], g['faces'], GRIPPER_SUBDIVISIONS); Ve, Fe = _synth.subdivide(g['enclosed_vertices'], g['enclosed_faces'], GRIPPER_SUBDIVISIONS)
        n_same = 96
        t0 = time.perf_counter()
        co.filter_grasp_pose(can[:max(1, n_same // (2 * len(sym)))], sym, batch.nocs_pose[0], I4, I4, I4, g['gripper_in_grasp'], 1, 0, 1, V, F, Ve, Fe, ob['xyz'], bg, 0.0005)
        co.filter_grasp_pose(P[:n_same // 2], [I4], I4, I4, I4, I4, g['gripper_in_grasp'], 1, 0, 0, V, F, Ve, Fe, ob['xyz'], bg, 0.0005)
        t_coll_same = (time.perf_counter() - t0) / (max(1, n_same // (2 * len(sym))) * len(sym) + n_same // 2)
    finally:
        co.lib().cr_set_variant(0, ctypes_float0(), 0)       # the oracle's default predicate (float64 clipping) for whoever uses it next
    pss = oref.prepared_state_dict(sd_seg)
    t0 = time.perf_counter()
    ids = tref.draw_ids(len(ob['xyz']), 8192)
    xin = tref.nunocs_transform(ob['xyz'].copy(), ob['normal'].copy(), ids)['input']
    with torch.no_grad():
        lg, _ = oref.pointnet_seg_forward_nnops(pss, torch.from_numpy(xin[None]).float())
    tref.nunocs_decode(lg[0].numpy(), 100)
    t_nunocs_obj = time.perf_counter() - t0
    per_rank = batch.n_total
    per_cand = t_transform + t_netonly + t_coll + t_nunocs_obj * len(batch.objs) / per_rank
    return {'value': round(1.0 / per_cand, 2), 'unit': 'candidates/s', 'cores': nthreads, 'host_cores': host, 'kind': 'port',
            'kind_note': 'port pinned to reference goldens: the F.conv1d / F.batch_norm op sequence of oracle/pointnet_ref.py is checked against outputs of the '
                         'imported /root/reference/pointnet2.py (tests/golden/make_golden*.py); the reference package itself cannot travel to the GPU box',
            'collision_threads': co.num_threads(),
            'collision_ms_per_evaluation_on_the_steps_subdivided_meshes': None if t_coll_same is None else round(t_coll_same * 1e3, 3),
            'sample': f'{n_score} candidates x (3 warm-ups, median of 3): python transform loop + PointNetCls fp32 in chunks of 200 through '
                      f'F.conv1d/F.batch_norm/F.linear on {nthreads} torch threads (best of the scan); {n_can * len(sym) + n_coll // 2} '
                      f'evaluations collision-filtered by the C/OpenMP restatement on {co.num_threads()} threads (both call shapes, structure build '
                      f'included; un-subdivided 36 / 48-triangle meshes of the same gripper surfaces, float32 SAT: the restatement has no BVH); '
                      f'1 NUNOCS forward amortised over {per_rank} candidates',
            'net_only_ms_per_candidate': round(t_netonly * 1e3, 3), 'transform_ms_per_candidate': round(t_transform * 1e3, 3),
            'net_plus_transform_candidates_per_s': round(1.0 / (t_transform + t_netonly), 2),
            'collision_ms_per_evaluation': round(t_coll * 1e3, 5), 'nunocs_ms_per_object': round(t_nunocs_obj * 1e3, 1),
            'thread_scan_ms_per_candidate': scan}


PEAK_VALU_WAVE_INSTS_PER_S = 256 * 2.4e9                # one wave64 vector instruction per CU and clock (4 SIMD16 x 4 cycles), 256 CUs, 2.4 GHz


def pmc_filter_issue(args):
    """The issue-side counters of the filter's grid kernel for THIS run's scene: one child run of this script (--pmc-filter-child: the
    step's filter sequence, 4 rounds) under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY` (counters only).
    -> ({'valu_wave_insts', 'wave_cycles', 'wait_any'} per pair of calls, source text) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    tmp = tempfile.mkdtemp(prefix='cg_pmcf_', dir='/tmp')
    counters = ('SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY')
    try:
        cmd = [exe, '--pmc', *counters, '--kernel-include-regex', 'filter_grasp_pose_kernel', '--output-format', 'csv', '-d', tmp, '--',
               sys.executable, os.path.abspath(__file__), '--pmc-filter-child', '--gpus', '1', '--workload', args.workload,
               '--candidates', str(args.candidates), '--candidates-total', str(args.candidates_total)]
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return None, f'rocprofv3 exited with {r.returncode}: {r.stderr[-300:]}'
        child = [ln for ln in r.stdout.splitlines() if ln.startswith('{"pmc_filter_child"')]
        if not child:
            return None, 'the counter pass printed no child record'
        rounds = json.loads(child[-1])['rounds']
        tot = defaultdict(float)
        for path in glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row.get('Kernel_Name') or row.get('kernel_name') or ''
                    if 'filter_grasp_pose_kernel<true,' in name.replace('(anonymous namespace)::', '') or 'filter_grasp_pose_kernelILb1' in name:
                        tot[row.get('Counter_Name') or row.get('counter_name')] += float(row.get('Counter_Value') or row.get('counter_value'))
        if not all(tot.get(c) for c in counters):
            return None, f'counters missing in the rocprofv3 output: {dict(tot)}'
        return ({'valu_wave_insts': tot['SQ_INSTS_VALU'] / rounds, 'wave_cycles': tot['SQ_WAVE_CYCLES'] / rounds, 'wait_any': tot['SQ_WAIT_ANY'] / rounds},
                f'rocprofv3 --pmc {" ".join(counters)} over {rounds} rounds of the step\'s filter sequence (child pass of this command)')
    except Exception as e:
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(args, precision, workload=None, candidates_total=None):
    """roofline.traffic measured for THIS run's workload instead of quoted from profiles/ (opt-in: --pmc-traffic, N = 1): two child
    runs of this script under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only -- no tracing
    domain next to --pmc; MI355X_MICROARCH.md, HBM section), one warm-up + one step each; HBM bytes of the encoder-pass kernel =
    (2 x FETCH_SIZE [gfx950 correction] + WRITE_SIZE) x 1024 summed over its dispatches / the candidates those dispatches scored.
    -> (bytes per candidate, source text) or (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    pat = re.compile(r'pointmlp_max(_split)?_kernel<2[,>]')
    tmp = tempfile.mkdtemp(prefix='cg_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    kb, cand = {}, None
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__), '--pmc-child',
                   '--gpus', '1', '--precision', precision, '--workload', workload or args.workload, '--candidates', str(args.candidates),
                   '--candidates-total', str(candidates_total or args.candidates_total), '--steps', '1', '--warmup', '1']
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return None, f'rocprofv3 --pmc {counter} exited with {r.returncode}: {r.stderr[-300:]}'
            child = [ln for ln in r.stdout.splitlines() if ln.startswith('{"pmc_child"')]
            if not child:
                return None, 'the counter pass printed no child record'
            cand = json.loads(child[-1])['candidate_equivalents']
            total, rows = 0.0, 0
            for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        name = row.get('Kernel_Name') or row.get('kernel_name') or ''
                        if (row.get('Counter_Name') or row.get('counter_name')) == counter and pat.search(name):
                            total += float(row.get('Counter_Value') or row.get('counter_value')); rows += 1
            if rows == 0:
                return None, f'no {counter} rows for the encoder-pass kernel in the rocprofv3 output'
            kb[counter] = total
    except Exception as e:          # the measurement is an extra: never let it take the bench line down
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per = (2.0 * kb['FETCH_SIZE'] + kb['WRITE_SIZE']) * 1024.0 / cand
    return per, (f'measured for this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate child passes of this command, 1 warm-up + 1 step '
                 f'each): (2 x {kb["FETCH_SIZE"]:.0f} + {kb["WRITE_SIZE"]:.0f}) KB over {cand:.0f} candidates = {per:.0f} B/candidate, x candidates per launch')


def rccl_selftest(batch, n_total, ref_out, device):
    """The collective of the N-rank job on RCCL itself, on this one GPU (outside the timed region): a ONE-rank `nccl` process group, the
    step's records gathered with the collective forced (catgrasp_amd/distributed.py: pad -> all_gather_into_tensor -> trim) and compared
    with the records of the timed run.  Never takes the bench line down: a failure is reported in the block."""
    import socket
    from catgrasp_amd import distributed as cgd
    info = {'what': 'one-rank nccl group, all_gather_into_tensor forced on the step records', 'ok': False}
    try:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=device)
        try:
            with torch.no_grad():
                rec = batch.score_slice(0, n_total)
                per, _ = cgd.shard_bounds(n_total, 1)
                cgd.gather_records(rec[:64], 64, 64, force_collective=True)        # communicator set-up outside the timing
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = cgd.gather_records(rec, per, n_total, force_collective=True)
                e1.record(); torch.cuda.synchronize()
            info.update(ok=bool(torch.equal(out, rec) and torch.equal(out, ref_out)), backend=torch.distributed.get_backend(),
                        rccl_version='.'.join(str(v) for v in torch.cuda.nccl.version()), records=int(n_total), bytes=int(out.numel() * 4),
                        all_gather_ms=round(e0.elapsed_time(e1), 4))
        finally:
            torch.distributed.destroy_process_group()
    except Exception as e:
        info['error'] = f'{type(e).__name__}: {e}'[:300]
    return info


_REAL_STDOUT = None


def claim_stdout():
    """stdout is the driver's channel for ONE JSON line.  Libraries under this process write there too (RCCL prints a version banner
    through C stdio when a communicator is created): point file descriptor 1 at stderr for the life of the process and keep the
    original for emit()."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                    help='strong (default): one fixed batch cut over the ranks; weak (C3 only): --candidates per GPU')
    ap.add_argument('--workload', choices=['C3', 'C4', 'C5'], default='C3', help='BASELINE.json configs[2] / [3] / [4]')
    ap.add_argument('--candidates', type=int, default=50000,
                    help='C3: grasp candidates per step -- in total (strong, the default) or per GPU (--scaling weak)')
    ap.add_argument('--candidates-total', type=int, default=None,
                    help='strong scaling: candidates per step over ALL GPUs (default C4: 200,000; C5: 500,000)')
    ap.add_argument('--precision', choices=['f32', 'f16x3', 'bf16x3', 'f16fp8x2'], default=None,
                    help='arithmetic of the timed path (`value`): f32 = exact-f32 MFMA (the reference\'s arithmetic); f16x3 / bf16x3 = split MFMA '
                         'products (3 MFMAs on hi+lo 16-bit pieces, f32 accumulation; logits within ~2e-6 / ~2e-5 of the float64 evaluation).  '
                         'Default f32; bf16x3 for C5 (configs[4] names the bf16 MFMA path)')
    ap.add_argument('--secondary', default='f16x3,bf16x3,f16fp8x2', help='comma list of further precisions measured in the same run ("" = none)')
    ap.add_argument('--chunk', type=int, default=None,
                    help='candidates per network launch (default: GraspPredicter\'s 16,384 = 805 MB of materialised input per chunk)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-api', action='store_true', help='skip the reference-API wall-clock block')
    ap.add_argument('--pmc-traffic', dest='pmc_traffic', action='store_true', default=None,
                    help='measure roofline.traffic for this run with two rocprofv3 --pmc child passes (N = 1 only; adds ~1 min).  Default: on '
                         'when rocprofv3 is present')
    ap.add_argument('--no-pmc-traffic', dest='pmc_traffic', action='store_false', help='quote the constant from profiles/ instead')
    ap.add_argument('--pmc-traffic-all', action='store_true', help='measure the traffic of every secondary precision too (two child passes each)')
    ap.add_argument('--no-projection', action='store_true', help='skip the projected_scaling block (N = 1: slice timings of the 2 / 4 / 8-rank shards)')
    ap.add_argument('--no-configs', action='store_true', help='skip the `configs` block of the default line (C4 and C5 at full size, 2 warm-ups + 3 steps each)')
    ap.add_argument('--no-rccl-selftest', action='store_true', help='skip the one-rank RCCL all-gather check after the timed region (N = 1)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)      # the child run of --pmc-traffic: workload only
    ap.add_argument('--pmc-filter-child', action='store_true', help=argparse.SUPPRESS)   # the child run of pmc_filter_issue: object 0's filter calls
    args = ap.parse_args()
    if args.scaling == 'weak' and args.workload != 'C3':
        ap.error(f'--workload {args.workload} is a strong-scaling workload')
    if args.pmc_traffic is None:
        import shutil
        args.pmc_traffic = bool(shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3'))
    claim_stdout()
    if args.precision is None:
        args.precision = 'bf16x3' if args.workload == 'C5' else 'f32'
    if args.candidates_total is None:
        args.candidates_total = 500000 if args.workload == 'C5' else 200000

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device')
    # dev only: CATGRASP_BENCH_BACKEND=gloo CATGRASP_BENCH_DEVICE=0 runs N ranks on ONE GPU to check the multi-rank control flow
    backend = os.environ.get('CATGRASP_BENCH_BACKEND', 'nccl')
    local_rank = int(os.environ.get('CATGRASP_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # N > 1: a census of what the process group really is -- how many ranks the collective library saw (an all-reduce of ones), and per
    # rank the device it computes on (index, name, PCI bus id, which peers it can reach directly) -- so that a SCALE record proves N
    # ranks on N distinct GPUs were in the job.  Gathered once, outside any timed region.
    rccl_info = None
    if world > 1:
        ones = torch.ones((1,), dtype=torch.float32, device=device if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(ones)
        prop = torch.cuda.get_device_properties(local_rank)
        mine = {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'current_device': int(torch.cuda.current_device()), 'name': prop.name,
                'pci_bus_id': getattr(prop, 'pci_bus_id', None), 'uuid': str(getattr(prop, 'uuid', '')),
                'peers_reachable': [j for j in range(torch.cuda.device_count()) if j != local_rank and torch.cuda.can_device_access_peer(local_rank, j)],
                'visible_devices': torch.cuda.device_count(), 'HIP_VISIBLE_DEVICES': os.environ.get('HIP_VISIBLE_DEVICES'),
                'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, mine)
        rccl_info = {'backend': backend, 'ranks_seen': int(round(float(ones.item()))), 'world_size': world,
                     'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if backend == 'nccl' else None,
                     'distinct_devices': len({(r['current_device'], r['pci_bus_id'], r['uuid']) for r in ranks}), 'ranks': ranks}

    from catgrasp_amd import distributed as cgd
    from catgrasp_amd import engine, ops, synth
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    from catgrasp_amd.workload import SceneBatch
    gp_kw = {'chunk': args.chunk} if args.chunk else {}
    clock = {'start': time.perf_counter()}
    timing = {}

    def lap(name):
        now = time.perf_counter()
        timing[name] = round(timing.get(name, 0.0) + now - clock['start'], 2)
        clock['start'] = now

    def build_workload(workload, scaling='strong', n=None):
        """-> dict(batch, n_total, cats, sds, gps, npreds).  One GraspPredicter / NunocsPredicter per category, each with its own seeded
        random-init weights (run_grasp_simulation.py:701-702).  strong: one fixed batch cut into `world` contiguous slices (C3: --candidates
        in total; C4 / C5: --candidates-total); weak (C3): every rank scores its own replica of the candidate set (global order is
        replica-major: slice r == replica r).  A rank only generates the candidate poses of its own slice."""
        cats = {'C3': ['nut'], 'C4': ['screw'], 'C5': ['nut', 'hnm', 'screw']}[workload]
        sds = {c: (synth.make_state_dict('cls', 6, 10, seed=2 * i), synth.make_state_dict('seg', 6, 300, seed=2 * i + 1)) for i, c in enumerate(cats)}
        gps = {c: GraspPredicter(c, cfg=DEFAULT_GRASP_CFG, state_dict=sds[c][0], device=device, **gp_kw) for c in cats}
        npreds = {c: NunocsPredicter(c, cfg=DEFAULT_NUNOCS_CFG, state_dict=sds[c][1], device=device) for c in cats}
        if scaling == 'weak':
            n = args.candidates * world
            b = SceneBatch(device, gps, npreds, kind='nut', n_objects=8, pts_per_object=2500, per_replica=args.candidates, replicas=world,
                           materialize=(rank * args.candidates, (rank + 1) * args.candidates), gripper_subdivisions=GRIPPER_SUBDIVISIONS)
        else:
            if n is None and workload == 'C3':
                n = args.candidates
            elif n is None:
                n = args.candidates_total if workload == args.workload else {'C4': 200000, 'C5': 500000}[workload]
            _, bounds = cgd.shard_bounds(n, world)
            kind, n_obj = {'C3': ('nut', 8), 'C4': ('screw', 16), 'C5': ('bin', 24)}[workload]
            b = SceneBatch(device, gps, npreds, kind=kind, n_objects=n_obj, pts_per_object=2500, per_replica=n, replicas=1, materialize=bounds[rank],
                           gripper_subdivisions=GRIPPER_SUBDIVISIONS)
        assert b.n_total == n
        return {'batch': b, 'n_total': n, 'cats': cats, 'sds': sds, 'gps': gps, 'npreds': npreds}

    wl = build_workload(args.workload, args.scaling)
    batch, n_total, cats, gps, npreds = wl['batch'], wl['n_total'], wl['cats'], wl['gps'], wl['npreds']
    sd_cls, sd_seg = wl['sds'][cats[0]]
    gp = gps[cats[0]]
    lap('build the workload (scene, candidates, predicters)')

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.pmc_filter_child:   # counter pass of pmc_filter_issue: the step's filter sequence, 4 rounds
        rects = step_filter_rects(batch)
        rounds = 4
        for _ in range(rounds):
            batch.run_filter_many(('roofline', 0, batch.n_total), rects)
        torch.cuda.synchronize()
        emit({'pmc_filter_child': True, 'rounds': rounds})
        return

    if args.pmc_child:          # counter pass of --pmc-traffic: run the step, report how many candidates the encoder-pass kernel scored
        engine.set_precision(args.precision)
        ops.KERNEL_TIMER = {'mid_mode': 2, 'events': []}
        with torch.no_grad():
            for _ in range(args.warmup + args.steps):
                cgd.score_sharded(batch.score_slice, n_total)
        torch.cuda.synchronize()
        ev = ops.KERNEL_TIMER['events']
        emit({'pmc_child': True, 'launches': len(ev), 'candidate_equivalents': float(sum(B * N / 2048.0 for _, _, (B, N) in ev))})
        return

    def measure(precision, batch=batch, n_total=n_total, steps=None, warmup=None):
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        engine.set_precision(precision)
        marks = []
        with torch.no_grad():
            for _ in range(warmup):
                cgd.score_sharded(batch.score_slice, n_total)
            barrier()
            ops.KERNEL_TIMER = {'mid_mode': 2, 'events': [], 'bgi_events': []}
            t0 = time.perf_counter()
            for _ in range(steps):
                out = cgd.score_sharded(batch.score_slice, n_total, marks=marks)
            barrier()
            dt_local = time.perf_counter() - t0
            timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
        t = torch.tensor([dt_local], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        score_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in marks]))
        gather_ms = float(np.mean([b.elapsed_time(c) for _, b, c in marks]))
        per_rank = [score_ms, gather_ms, dt_local / steps * 1e3]
        if world > 1:
            lst = [None] * world
            torch.distributed.all_gather_object(lst, per_rank)
            per_rank = lst
        else:
            per_rank = [per_rank]
        ev = timer['events']
        k_ms = [a.elapsed_time(b) for a, b, _ in ev]
        cand = [B * N / 2048.0 for _, _, (B, N) in ev]
        res = {'dt': dt, 'out': out, 'launches': len(k_ms), 'per_rank': per_rank,
               'avg_ms': float(np.mean(k_ms)) if k_ms else float('nan'), 'avg_cand': float(np.mean(cand)) if cand else 0.0}
        res['tflops'] = 2.0 * MAC_PER_POINT_ENC * 2048 * res['avg_cand'] / (res['avg_ms'] * 1e-3) / 1e12 if k_ms else float('nan')
        bgi = timer['bgi_events']
        if bgi:
            b_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in bgi]))
            b_bytes = float(np.mean([g * (2048 * (4 + 24) + 48) for _, _, g in bgi])) + batch.cloud_xyz.numel() * 8
            res['bgi'] = {'bound': 'hbm', 'kernel': 'build_grasp_input_staged_kernel (gather resampled points + pose transform; 24 B written + 4 B id read per point)',
                          'achieved': round(b_bytes / (b_ms * 1e-3) / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                          'frac': round(b_bytes / (b_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), 'avg_launch_ms': round(b_ms, 4), 'launches': len(bgi),
                          'algorithmic_bytes_per_launch': int(b_bytes)}
        return res

    measured_traffic = {}

    def roofline(precision, r, traffic=None):
        # traffic is reported only when it was MEASURED for this run (primary precision, N = 1, rocprofv3 present); otherwise null plus
        # the round-2 PMC constant under its own name -- not a counter read in this run
        per, source = None, ('not measured in this run; `traffic_round2_constant` = profiles/r2_pmc_hbm_pointmlp_{f32,split,f16fp8x2}.csv '
                             '(2*FETCH_SIZE + WRITE_SIZE per candidate at B=4096) x candidates per launch; algorithmic bytes are 69,632 B/candidate')
        if traffic is not None:
            per, source = traffic
        elif precision in measured_traffic:
            per, source = measured_traffic[precision]
        hbm_gbs = ALG_HBM_BYTES_PER_CANDIDATE * r['avg_cand'] / (r['avg_ms'] * 1e-3) / 1e9
        common = {'achieved': round(r['tflops'], 2), 'unit': 'TFLOP/s', 'avg_launch_ms': round(r['avg_ms'], 4), 'launches': r['launches'],
                  'candidates_per_launch': round(r['avg_cand'], 1), 'flop_per_candidate_launch': 2 * MAC_PER_POINT_ENC * 2048,
                  'traffic': int(per * r['avg_cand']) if per is not None else None,
                  'traffic_source': source,
                  'traffic_round2_constant': int(PMC_HBM_BYTES_PER_CANDIDATE[precision] * r['avg_cand']),
                  'hbm_achieved_gbs': round(hbm_gbs, 1), 'hbm_frac': round(hbm_gbs / PEAK_HBM_GBS, 5)}
        if precision == 'f32':
            return dict({'bound': 'mfma', 'kernel': 'pointmlp_max_kernel<2> (encoder pass: conv1, x.T64, conv2, conv3, max; exact-f32 MFMA)',
                         'peak': PEAK_F32_MFMA_TFLOPS, 'frac': round(r['tflops'] / PEAK_F32_MFMA_TFLOPS, 4)}, **common)
        if precision == 'f16fp8x2':
            units = 2.0 * L3_SHARE + 3.0 * (1.0 - L3_SHARE)          # matrix-pipe time per algorithmic flop, in 16-bit-MFMA flop equivalents
            return dict({'bound': 'mfma',
                         'kernel': 'pointmlp_max_split_kernel<2, 8, true, true> (encoder pass; 128->1024 layer = 1 f16 MFMA + 2 block-scaled e4m3 MFMAs '
                                   'at half cost each per product block; front layers 3 f16 MFMAs)',
                         'peak': PEAK_16BIT_MFMA_TFLOPS, 'frac': round(r['tflops'] / PEAK_16BIT_MFMA_TFLOPS, 4),
                         'issued_mfma_tflops': round(units * r['tflops'], 1), 'issued_frac': round(units * r['tflops'] / PEAK_16BIT_MFMA_TFLOPS, 4),
                         'note': f'achieved counts ALGORITHMIC flops; the matrix pipe spends {units:.3f} 16-bit-MFMA flop equivalents per algorithmic flop '
                                 '(an e4m3 MX MFMA does 2x the flops per pass), so the ceiling for algorithmic flops is peak/'
                                 f'{units:.3f} = {PEAK_16BIT_MFMA_TFLOPS / units:.0f} TFLOP/s'}, **common)
        el = 'f16' if precision == 'f16x3' else 'bf16'
        return dict({'bound': 'mfma',
                     'kernel': f'pointmlp_max_split_kernel<2, 8, {"true" if el == "f16" else "false"}> (encoder pass; 3 {el} MFMAs per algorithmic product block)',
                     'peak': PEAK_16BIT_MFMA_TFLOPS, 'frac': round(r['tflops'] / PEAK_16BIT_MFMA_TFLOPS, 4),
                     'issued_mfma_tflops': round(3 * r['tflops'], 1), 'issued_frac': round(3 * r['tflops'] / PEAK_16BIT_MFMA_TFLOPS, 4),
                     'note': 'achieved counts ALGORITHMIC flops; the split issues 3 16-bit MFMA flops per algorithmic flop, so the ceiling for '
                             'algorithmic flops is peak/3 = 833 TFLOP/s'}, **common)

    prim = measure(args.precision)
    ref_out = prim['out'].clone()
    lap('primary measurement (warm-up + timed steps)')
    # the reference-API wall-clock right behind the primary measurement (same clock / thermal state as `value`), before the minutes of
    # split-precision runs below
    api = api_block(batch, gp, device) if (rank == 0 and world == 1 and not args.no_api) else None
    if api is not None:
        try:
            api['pick_cycle'] = pick_cycle_block(batch, gp, npreds[cats[0]], device)
        except Exception as e:          # an extra: never let it take the bench line down
            api['pick_cycle'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    lap('api block (reference entry points, pick cycle)')
    secondary = []
    for other in [p for p in args.secondary.split(',') if p and p != args.precision]:
        r = measure(other)
        secondary.append({'precision': other, 'dtype': DTYPE[other], 'value': round(n_total * args.steps / r['dt'], 1),
                          'ms_per_step': round(r['dt'] / args.steps * 1e3, 3), 'roofline': roofline(other, r),
                          'max_abs_p_G_difference_vs_primary': float((r['out'][:, 0] - ref_out[:, 0]).abs().max().item()),
                          'codes_identical_to_primary': bool(torch.equal(r['out'][:, 1], ref_out[:, 1]))})
    if world > 1 and args.workload == 'C3' and args.scaling == 'strong':
        # the weak-scaling figure of the same scene next to the strong one: every rank scores its own --candidates (different seeds)
        ww = build_workload('C3', 'weak')
        wbatch, wn = ww['batch'], ww['n_total']
        r = measure(args.precision, wbatch, wn)
        secondary.append({'scaling': 'weak', 'precision': args.precision, 'dtype': DTYPE[args.precision], 'candidates_per_gpu': args.candidates,
                          'candidates_total': wn, 'value': round(wn * args.steps / r['dt'], 1), 'ms_per_step': round(r['dt'] / args.steps * 1e3, 3),
                          'per_rank_ms': [[round(v, 3) for v in pr] for pr in r['per_rank']],
                          'note': 'per-GPU work fixed: N x the single-GPU batch, gathered by the same one all_gather'})
        del wbatch, ww, r
    engine.set_precision(args.precision)
    lap('secondary precisions')

    projected = None
    if world == 1 and rank == 0 and args.scaling == 'strong' and not args.no_projection:
        projected = projected_scaling_block(batch, n_total, prim['dt'] / args.steps)
    selftest = None
    if world == 1 and not args.no_rccl_selftest and backend == 'nccl':
        selftest = rccl_selftest(batch, n_total, ref_out, device)
    lap('projected scaling + rccl self-test')

    if rank == 0 and world == 1 and args.pmc_traffic:
        # the primary arithmetic and, when it is among the secondaries, bf16x3 (the C5 arithmetic); --pmc-traffic-all: every secondary
        wanted = [args.precision] + [x['precision'] for x in secondary if 'precision' in x and 'roofline' in x
                                     and (args.pmc_traffic_all or x['precision'] == 'bf16x3')]
        for prec in wanted:
            per, why = pmc_traffic(args, prec)
            if per is not None:
                measured_traffic[prec] = (per, why + '; algorithmic bytes are 69,632 B/candidate')
            else:
                print(f'bench.py: --pmc-traffic failed for {prec} ({why}); quoting the constant from profiles/', file=sys.stderr)
        for x in secondary:                     # the secondaries' roofline blocks were built before their traffic was measured
            if x.get('precision') in measured_traffic and 'roofline' in x:
                per, source = measured_traffic[x['precision']]
                x['roofline']['traffic'] = int(per * x['roofline']['candidates_per_launch'])
                x['roofline']['traffic_source'] = source
    lap('traffic counter passes (rocprofv3 --pmc children)')
    METRIC = {'C3': 'grasp candidates scored+collision-checked /sec, 20k-pt clutter scene',
              'C4': 'grasp candidates scored+collision-checked /sec, 40k-pt scene, candidates sharded over the GPUs',
              'C5': 'grasp candidates scored+collision-checked /sec, 60k-pt mixed-category bin, candidates sharded over the GPUs'}

    def describe(workload, batch, n_total, cats, out, scaling='strong'):
        """The `config` object of a workload's line: what was run, on what, and the reject-code histogram of its records."""
        from catgrasp_amd.workload import SYMMETRY_COUNT
        sym_txt = ' / '.join(str(SYMMETRY_COUNT[c]) for c in cats)
        codes = out[:, 1].long()
        return {'workload': {'C3': 'C3 (BASELINE.json configs[2]): nut clutter pile, 20k-pt scene (8 objects x 2500 pts), ' +
                                   (f'{args.candidates} candidates/GPU' if scaling == 'weak' else
                                    f'{n_total} candidates in total over {world} GPU(s)'),
                             'C4': 'C4 (BASELINE.json configs[3]): screw category, 40k-pt scene (16 objects x 2500 pts), '
                                   f'{n_total} candidates in total over {world} GPU(s)',
                             'C5': 'C5 (BASELINE.json configs[4]): mixed-category bin, 60k-pt scene (24 objects x 2500 pts: nut / hnm / screw '
                                   'in turn, one GraspPredicter + NunocsPredicter per category), '
                                   f'{n_total} candidates in total over {world} GPU(s)'}[workload] +
                            f': NUNOCS PointNetSeg(8192x6) per object + filterGraspPose [{len(batch.gripper["faces"])} / {len(batch.gripper["enclosed_faces"])}-triangle '
                            f'gripper meshes; canonical grasps x {sym_txt} symmetries with '
                            'adjust_collision_pose=True (grasp_sampler.py:345) and cone poses with symmetry=[I] (grasp_sampler.py:216)] + '
                            'device pose inverse + per-candidate resampling draw + grasp-Q PointNetCls(2048x6) + softmax/p_G for EVERY candidate',
                'candidates_per_gpu': n_total // world, 'candidates_total': n_total, 'scene_points': int(batch.cloud_xyz.shape[0]),
                'symmetries': {c: SYMMETRY_COUNT[c] for c in cats} if len(cats) > 1 else batch.n_sym,
                'evaluations_nocs_shape_adjust_true': int(sum(s.count for s in batch.segs if s.kind == 'nocs')),
                'evaluations_cone_shape_adjust_false': int(sum(s.count for s in batch.segs if s.kind == 'cone')),
                'reject_code_histogram_0keep_1dir_2ik_3open_4enclosed': torch.bincount(codes, minlength=5).tolist(),
                'parallelism': f'candidate-shard x{world} ({scaling})'}

    def config_block(workload, precision, steps=3, warmup=2, pmc_total=None):
        """One further BASELINE.json configuration measured in this same process (the default N = 1 run carries C4 and C5 under `configs`):
        the workload of `--workload <workload>` at its full size, `warmup` + `steps` steps, the dominant kernel's roofline from its own
        HIP events, the projected 2 / 4 / 8-rank speed-ups from slice timings, and -- when rocprofv3 is on the box -- `traffic` from
        two counter passes of the same workload at `pmc_total` candidates (bytes per candidate x the candidates per launch of the
        full-size run: the kernel's traffic per candidate does not depend on how many launches the job has)."""
        import hashlib
        t0 = time.perf_counter()
        w = build_workload(workload)
        b, n = w['batch'], w['n_total']
        t1 = time.perf_counter()
        r = measure(precision, b, n, steps=steps, warmup=warmup)
        t2 = time.perf_counter()
        blk = {'metric': METRIC[workload], 'value': round(n * steps / r['dt'], 1), 'unit': 'candidates/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
               'ms_per_step': round(r['dt'] / steps * 1e3, 3), 'scaling': 'strong', 'dtype': DTYPE[precision],
               'config': describe(workload, b, n, w['cats'], r['out']),
               'records_sha256': hashlib.sha256(r['out'].cpu().numpy().tobytes()).hexdigest()}
        traffic = None
        if args.pmc_traffic and pmc_total:
            per, why = pmc_traffic(args, precision, workload=workload, candidates_total=pmc_total)
            if per is not None:
                traffic = (per, why + f' [counter passes at {pmc_total} candidates of this workload]; algorithmic bytes are 69,632 B/candidate')
            else:
                print(f'bench.py: counter passes for {workload} failed ({why})', file=sys.stderr)
        blk['roofline'] = roofline(precision, r, traffic=traffic)
        t3 = time.perf_counter()
        if not args.no_projection:
            blk['projected_scaling'] = projected_scaling_block(b, n, r['dt'] / steps, steps=1)
        engine.set_precision(args.precision)
        t4 = time.perf_counter()
        blk['wall_s'] = {'build': round(t1 - t0, 2), 'warm-up + steps': round(t2 - t1, 2), 'traffic counter passes': round(t3 - t2, 2),
                         'projected scaling': round(t4 - t3, 2), 'total': round(t4 - t0, 2)}
        return blk

    if rank == 0:
        dt = prim['dt']
        line = {
            'metric': METRIC[args.workload],
            'value': round(n_total * args.steps / dt, 1), 'unit': 'candidates/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': DTYPE[args.precision],
            'data': 'synthetic (seeded clouds/candidates/gripper, random-init weights)',
            'config': describe(args.workload, batch, n_total, cats, ref_out, args.scaling),
            'roofline': roofline(args.precision, prim),
            'per_rank_ms': {'columns': ['local scoring (HIP events)', 'all_gather of the (p_G, code) records (HIP events)', 'step wall-clock'],
                            'ranks': [[round(v, 3) for v in pr] for pr in prim['per_rank']]},
        }
        import hashlib
        line['records_sha256'] = hashlib.sha256(ref_out.cpu().numpy().tobytes()).hexdigest()      # (p_G, code) of every candidate, global order
        if rccl_info is not None:
            line['rccl'] = rccl_info
        if projected is not None:
            line['projected_scaling'] = projected
        if selftest is not None:
            line['rccl_selftest'] = selftest
        if 'bgi' in prim:
            line['roofline_hbm'] = prim['bgi']
        if world == 1:
            try:
                line['roofline_filter'] = filter_roofline_block(batch, device, issue=pmc_filter_issue(args) if args.pmc_traffic else (None, '--no-pmc-traffic'))
            except Exception as e:          # an extra: never let it take the bench line down
                line['roofline_filter'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        if secondary:
            line['secondary'] = secondary
        if api is not None:
            line['api'] = api
            try:
                line['pp_encoder'] = encoder_block(batch, device)
            except Exception as e:          # an extra: never let it take the bench line down
                line['pp_encoder'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        lap('filter roofline, encoder block')
        if world == 1 and args.workload == 'C3' and args.scaling == 'strong' and not args.no_configs:
            # the other two GPU configurations of BASELINE.json in the same driver-run line
            line['configs'] = {}
            for wl_name, prec, pmc_total in (('C4', 'f32', 50000), ('C5', 'bf16x3', 150000)):
                try:
                    line['configs'][wl_name] = config_block(wl_name, prec, pmc_total=pmc_total)
                except Exception as e:      # an extra: never let it take the bench line down
                    line['configs'][wl_name] = {'error': f'{type(e).__name__}: {e}'[:300]}
                torch.cuda.empty_cache()
                lap(f'configs.{wl_name}')
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(batch, sd_cls, sd_seg)
            lap('cpu baseline')
        line['timing_s'] = dict(timing, total=round(sum(timing.values()), 2))
        emit(line)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
