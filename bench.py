#!/usr/bin/env python
"""bench.py -- grasp candidates scored + collision-checked per second (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
  NUNOCS net over the scene's objects  ->  filterGraspPose over the rank's candidates  ->
  grasp-Q net (input transform + PointNetCls + softmax + p_G) over the same candidates
  [-> one RCCL all_gather of the packed (p_G, code) records when --gpus > 1].
Workload (config.workload): BASELINE.json configs[1] scale -- nut clutter pile, 20k-pt scene
(8 objects x 2500 pts), 10k candidates per GPU, fp32 -- with the configs[2] stages (NUNOCS + collision)
included because the metric is "scored + collision-checked".  All inputs are resident in HBM before
the timed region; weights are seeded random (the reference ships no checkpoints), data synthetic.

Launch: python bench.py --gpus 1 --steps K --warmup W
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_PER_POINT_ENC = 9 + 384 + 4096 + 8192 + 131072      # encoder pass (mid_mode 2): T3, conv1, .T64, conv2, conv3
PEAK_F32_MFMA_TFLOPS = 157.3                            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0                          # MI355X_MICROARCH.md: BF16/F16 MFMA dense peak (~2.5 PF)
# HBM bytes per candidate of the encoder-pass kernel (mid_mode 2) of each arithmetic from the PMC passes in profiles/r1_pmc_pointmlp.csv
# (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, B=4096): 2*FETCH_SIZE (gfx950 correction for wide
# coalesced reads, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, divided by 4096.  Algorithmic: 49152 B x + 16384 B transform
# + 4096 B out = 69632 B/candidate; both kernels move the algorithmic bytes and nothing else (no scratch).
PMC_HBM_BYTES_PER_CANDIDATE = {'bf16x3': (2 * 133760.3 + 16384.0) * 1024 / 4096, 'f16x3': (2 * 133762.0 + 16384.0) * 1024 / 4096,
                               'f32': (2 * 133763.3 + 16384.0) * 1024 / 4096}


def build_workload(device, G, seed, n_objects=8, pts_per_object=2500, kind='nut'):
    from catgrasp_amd import my_cpp, synth, transforms
    objs = synth.make_scene(n_objects, pts_per_object, seed=0, kind=kind)           # same scene on every rank
    gripper = synth.make_gripper()
    rng = np.random.default_rng(1000 + seed)
    per = [G // n_objects + (1 if k < G % n_objects else 0) for k in range(n_objects)]
    clouds, offsets, pose_rows, poses_dev, scenes, ids = [], [], [], [], [], []
    off = 0
    gen = torch.Generator(device=device); gen.manual_seed(1234 + seed)
    for k, ob in enumerate(objs):
        dc = transforms.DeviceCloud(ob['xyz'], ob['normal'], device)
        clouds.append(dc); offsets.append(off)
        P = synth.make_candidates(ob, per[k], rng, gripper['hand_depth'], gripper['init_bite'])
        pose_rows.append(transforms.pose_inverse_rows(P, dc.center))
        poses_dev.append(torch.from_numpy(P.astype(np.float32).reshape(-1, 16)).to(device))
        bg = synth.background_points(objs, k, gripper['diameter'])
        scenes.append(my_cpp.GripperScene(gripper['vertices'], gripper['faces'], gripper['enclosed_vertices'],
                                          gripper['enclosed_faces'], ob['xyz'], bg, 0.0005, device))
        ids.append(transforms.draw_ids_device(dc.n, 2048, per[k], device, gen) + off)
        off += dc.n
    wl = {
        'objs': objs, 'gripper': gripper, 'per': per, 'scenes': scenes, 'poses_dev': poses_dev,
        'cloud_xyz': torch.cat([c.xyz for c in clouds]).contiguous(),
        'cloud_normal': torch.cat([c.normal for c in clouds]).contiguous(),
        'ids': torch.cat(ids).contiguous(),
        'pose_inv': torch.from_numpy(np.concatenate(pose_rows)).to(device),
        'nunocs_ids': torch.stack([transforms.draw_ids_device(c.n, 8192, 1, device, gen)[0] + o
                                   for c, o in zip(clouds, offsets)]).contiguous(),
        'G': G,
    }
    return wl


def run_step(wl, gp, npred, gather_buf=None, world=1):
    from catgrasp_amd import my_cpp
    I4 = np.eye(4, dtype=np.float32)
    # (1) NUNOCS canonicaliser over every object cloud of the scene
    coords, conf, _ = npred.nocs_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['nunocs_ids'])
    # (2) collision filter (cone-sampler call shape: symmetry=[I], nocs_pose=I, approach-dir filter on).
    #     One call per object as in the reference; the per-object kernels are small (1250 wavefronts each), so they are
    #     issued on separate HIP streams and run concurrently, then joined back into the main stream.
    codes = []
    sym = wl.setdefault('_sym', torch.eye(4, device=wl['cloud_xyz'].device).reshape(1, 16).contiguous())
    side = wl.setdefault('_streams', [torch.cuda.Stream(device=wl['cloud_xyz'].device) for _ in range(len(wl['scenes']))])
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main)
    for k, sc in enumerate(wl['scenes']):
        with torch.cuda.stream(side[k]):
            side[k].wait_event(fork)
            c, _, _ = my_cpp.filter_on_device(sc, wl['poses_dev'][k], sym, I4, I4, I4, I4, wl['gripper']['gripper_in_grasp'],
                                              True, False, False)
        codes.append(c)
    for st in side:
        main.wait_stream(st)
    codes = torch.cat(codes)
    # (3) grasp-Q scoring of every candidate
    probs, label, conf_q, p_g = gp.score_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['ids'], wl['pose_inv'])
    # (4) packed per-candidate record: p_G (f32) + code (as f32 lane) -> one all_gather over xGMI
    rec = torch.stack([p_g, codes.float()], dim=1).contiguous()
    if world > 1:
        if torch.distributed.get_backend() == 'nccl':
            torch.distributed.all_gather_into_tensor(gather_buf, rec)
        else:           # dev only (CATGRASP_BENCH_BACKEND=gloo): exercise the multi-rank control flow on a 1-GPU box
            parts = [torch.empty(rec.shape, dtype=rec.dtype) for _ in range(world)]
            torch.distributed.all_gather(parts, rec.cpu())
            gather_buf.copy_(torch.cat(parts))
        return gather_buf
    return rec


def cpu_baseline(wl, sd_cls, sd_seg, n_score=400, n_coll=4096):
    """The CPU oracle (a port of the reference path, oracle/) timed on this box's host cores on a bounded
    sample: python GraspDataset.transform loop + PointNetCls fp32 forward in chunks of 200
    (predicter.py:67-91), the C/OpenMP filterGraspPose restatement, and the NUNOCS forward amortised."""
    from oracle import collision_oracle as co
    from oracle import pointnet_ref as oref
    from oracle import transforms_ref as tref
    from catgrasp_amd import synth
    nthreads = min(os.cpu_count(), 16)      # best of an 8..256-thread scan on the 256-core bench host (oversubscription hurts)
    torch.set_num_threads(nthreads)
    ob = wl['objs'][0]; g = wl['gripper']
    rng = np.random.default_rng(7)
    P = synth.make_candidates(ob, max(n_score, n_coll), rng, g['hand_depth'], g['init_bite'])
    with torch.no_grad():
        oref.pointnet_cls_forward(sd_cls, torch.zeros(8, 2048, 6))     # warm-up (thread pool, allocator)
    t0 = time.time()
    xs = []
    for i in range(n_score):
        ids = tref.draw_ids(len(ob['xyz']), 2048)
        xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids)['input'])
    x = torch.from_numpy(np.stack(xs)).float()
    with torch.no_grad():
        logits = torch.cat([oref.pointnet_cls_forward(sd_cls, x[s:s + 200])[0] for s in range(0, n_score, 200)])   # predicter.py:69 batch 200
    tref.predict_batch_post(logits.numpy())
    t_score = (time.time() - t0) / n_score
    bg = synth.background_points(wl['objs'], 0, g['diameter'])
    I4 = np.eye(4)
    t0 = time.time()
    co.filter_grasp_pose(P[:n_coll], [I4], I4, I4, I4, I4, g['gripper_in_grasp'], 1, 0, 0, g['vertices'], g['faces'],
                         g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
    t_coll = (time.time() - t0) / n_coll
    t0 = time.time()
    ids = tref.draw_ids(len(ob['xyz']), 8192)
    xin = tref.nunocs_transform(ob['xyz'].copy(), ob['normal'].copy(), ids)['input']
    with torch.no_grad():
        lg, _ = oref.pointnet_seg_forward(sd_seg, torch.from_numpy(xin[None]).float())
    tref.nunocs_decode(lg[0].numpy(), 100)
    t_nunocs = (time.time() - t0) * len(wl['objs']) / wl['G']
    per_cand = t_score + t_coll + t_nunocs
    return {'value': round(1.0 / per_cand, 2), 'unit': 'candidates/s', 'cores': nthreads, 'host_cores': os.cpu_count(), 'kind': 'port',
            'sample': f'{n_score} candidates scored (python transform loop + fp32 torch oracle, {nthreads} threads) '
                      f'+ {n_coll} candidates collision-filtered (C/OpenMP oracle) + 1 NUNOCS forward amortised over {wl["G"]}',
            'score_ms_per_candidate': round(t_score * 1e3, 3), 'collision_ms_per_candidate': round(t_coll * 1e3, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--candidates', type=int, default=10000, help='grasp candidates per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['f16x3', 'bf16x3', 'f32'], default='f16x3',
                    help='arithmetic of the timed path: f16x3 / bf16x3 = split MFMA products (x = hi + lo half / bf16 pieces, 3 MFMAs per '
                         'product block, f32 accumulation; logits within ~2e-6 / ~2e-5 of the float64 evaluation; bar 1e-4), f32 = exact-f32 MFMA')
    ap.add_argument('--no-secondary', action='store_true', help='skip the second measurement with the other precision')
    args = ap.parse_args()

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

    from catgrasp_amd import ops, synth
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    sd_cls = synth.make_state_dict('cls', 6, 10, seed=0)
    sd_seg = synth.make_state_dict('seg', 6, 300, seed=1)
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd_cls, device=device)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=sd_seg, device=device)
    G = args.candidates
    wl = build_workload(device, G, seed=rank)
    gather_buf = torch.empty((world * G, 2), dtype=torch.float32, device=device) if world > 1 else None

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    from catgrasp_amd import engine

    def measure(precision):
        engine.set_precision(precision)
        with torch.no_grad():
            for _ in range(args.warmup):
                run_step(wl, gp, npred, gather_buf, world)
            barrier()
            ops.KERNEL_TIMER = {'mid_mode': 2, 'events': []}
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = run_step(wl, gp, npred, gather_buf, world)
            barrier()
            dt = time.perf_counter() - t0
            timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        ev = timer['events']
        k_ms = [a.elapsed_time(b) for a, b, _ in ev]
        k_flops = [2.0 * MAC_PER_POINT_ENC * B * N for _, _, (B, N) in ev]
        avg_ms = float(np.mean(k_ms)) if k_ms else float('nan')
        achieved = float(np.mean(k_flops)) / (avg_ms * 1e-3) / 1e12 if k_ms else float('nan')
        avg_B = float(np.mean([B * N / 2048.0 for _, _, (B, N) in ev])) if ev else None
        return dt, avg_ms, achieved, len(k_ms), out, avg_B

    def roofline(precision, achieved, avg_ms, n, avg_B=None):
        def traffic(prec):
            per = PMC_HBM_BYTES_PER_CANDIDATE.get(prec)
            return None if (per is None or avg_B is None) else int(per * avg_B)
        if precision == 'f32':
            return {'bound': 'mfma', 'kernel': 'pointmlp_max_kernel<2> (encoder pass: conv1, x.T64, conv2, conv3, max; exact-f32 MFMA)',
                    'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic('f32'), 'traffic_unit': 'HBM bytes per launch (PMC)',
                    'avg_launch_ms': round(avg_ms, 4),
                    'launches': n, 'flop_per_candidate_launch': 2 * MAC_PER_POINT_ENC * 2048}
        el = 'f16' if precision == 'f16x3' else 'bf16'
        return {'bound': 'mfma', 'kernel': f'pointmlp_max_split_kernel<2, 8, {"true" if el == "f16" else "false"}> (encoder pass; 3 {el} MFMAs per algorithmic product block)',
                'achieved': round(achieved, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), 'traffic': traffic(precision), 'traffic_unit': 'HBM bytes per launch (PMC)',
                'avg_launch_ms': round(avg_ms, 4),
                'launches': n, 'flop_per_candidate_launch': 2 * MAC_PER_POINT_ENC * 2048,
                'issued_mfma_tflops': round(3 * achieved, 1), 'issued_frac': round(3 * achieved / PEAK_BF16_MFMA_TFLOPS, 4),
                'note': 'achieved counts ALGORITHMIC flops; the split issues 3 16-bit MFMA flops per algorithmic flop, '
                        'so the ceiling for algorithmic flops is peak/3 = 833 TFLOP/s (f16 and bf16 MFMA share the 2.5 PFLOP/s dense peak)'}

    dt, avg_ms, achieved, n_launch, out, avg_B = measure(args.precision)
    secondary = None
    if not args.no_secondary:
        other = 'f32' if args.precision != 'f32' else 'f16x3'
        ref_out = out.clone()
        dt2, avg2, ach2, n2, out2, avg_B2 = measure(other)
        pg_diff = float((out2[:, 0] - ref_out[:, 0]).abs().max().item())
        secondary = {'precision': other, 'value': round(world * G * args.steps / dt2, 1), 'ms_per_step': round(dt2 / args.steps * 1e3, 3),
                     'roofline': roofline(other, ach2, avg2, n2, avg_B2), 'max_abs_p_G_difference_between_precisions': pg_diff,
                     'codes_identical': bool(torch.equal(out2[:, 1], ref_out[:, 1]))}
        engine.set_precision(args.precision)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * G * args.steps / dt
        line = {
            'metric': 'grasp candidates scored+collision-checked /sec, 20k-pt clutter scene',
            'value': round(value, 1), 'unit': 'candidates/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'f32' else
                     f'f32 in/out/accumulate; wide-layer products as 3x {"f16" if args.precision == "f16x3" else "bf16"} MFMA ({args.precision} split)',
            'data': 'synthetic (seeded clouds/candidates/gripper, random-init weights)',
            'config': {'workload': 'nut clutter pile, 20k-pt scene (8 objects x 2500 pts), '
                                   f'{G} candidates/GPU: NUNOCS(8x8192) + filterGraspPose + grasp-Q PointNetCls(2048x6)',
                       'candidates_per_gpu': G, 'scene_points': int(wl['cloud_xyz'].shape[0]), 'parallelism': f'candidate-shard x{world}'},
            'roofline': roofline(args.precision, achieved, avg_ms, n_launch, avg_B),
        }
        if secondary is not None:
            line['secondary'] = secondary
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(wl, sd_cls, sd_seg)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
