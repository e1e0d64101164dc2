"""TEST INFRASTRUCTURE ONLY (oracle) -- how much does the collision predicate depend on what could NOT be pinned to FCL / octomap?

The reference's `CollisionManager::isAnyCollision` is FCL's BVHModel<OBBRSSf> vs fcl::OcTree<float> (my_cpp/collision_manager.cpp:41-45,
63-70, 93-111); neither library is available here (PARITY UNPINNED, oracle/collision_ref.c).  This study runs the SAME filterGraspPose
control flow (common.cpp:156-321, both live call shapes, nudging on) over the WHOLE C3 batch of BASELINE.json configs[2] -- 50,000
evaluations, every object of the scene -- under alternative formulations of the leaf-box-vs-triangle predicate
(collision_ref.c: cr_set_variant) and counts what changes against the parity oracle (float64 polygon clipping, no separating axes):

  sat_f32       the 13-axis separating-axis test in float32 -- what csrc/collision.hip evaluates, operation for operation
  mpr_libccd    libccd's MPR intersection test on box / triangle support functions with FCL's defaults (mpr_tolerance 1e-6, double):
                the narrow phase FCL's default solver actually runs for a BVH-triangle / octree-leaf pair (restated, not linked)
  fcl_halving   leaf boxes from FCL's 16 float halvings of the root BV instead of ((float)k + 0.5f) * res
  fcl_halving+mpr_libccd   both: the closest this container can get to what the reference's FCL call computes
  grow_1um      box half edge + 1e-6 m   (libccd contact tolerance scale)
  shrink_1um    box half edge - 1e-6 m

plus 100,000 synthetic grazing triangle/box pairs per variant.  Output: profiles/r4_collision_sensitivity.json.
    python -m oracle.collision_sensitivity [--out profiles/r4_collision_sensitivity.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from catgrasp_amd import synth, transforms, workload      # noqa: E402  (host-side scene / candidate generators only: no device code)
from oracle import collision_oracle as co                 # noqa: E402

# (leaf_mode, half-edge delta, narrow phase: 0 float64 clipping = the parity oracle, 1 float32 SAT = the kernel's twin, 2 libccd MPR)
VARIANTS = {'sat_f32': (0, 0.0, 1), 'mpr_libccd': (0, 0.0, 2), 'fcl_halving': (1, 0.0, 0), 'fcl_halving+mpr_libccd': (1, 0.0, 2),
            'fcl_halving+sat_f32': (1, 0.0, 1), 'grow_1um': (0, 1e-6, 0), 'shrink_1um': (0, -1e-6, 0)}
NARROW_NAME = {0: 'clip64', 1: 'sat_f32', 2: 'mpr_libccd'}
I4 = np.eye(4)


def set_variant(leaf_mode=0, dh=0.0, narrow=0):
    co.lib().cr_set_variant(ctypes.c_int(leaf_mode), ctypes.c_float(dh), ctypes.c_int(narrow))


def c3_batch(n_objects=8, pts_per_object=2500, per_replica=50000, kind='nut'):
    """The C3 evaluations exactly as bench.py / SceneBatch plan them: [(segment, poses (n,4,4) f64)], scene objects, gripper."""
    objs = synth.make_scene(n_objects, pts_per_object, seed=0, kind=kind)
    gripper = synth.make_gripper()
    cats = [ob['kind'] for ob in objs]
    segs, n_total = workload.plan_segments(n_objects, per_replica, [workload.SYMMETRY_COUNT[c] for c in cats], 1)
    nocs = [workload.scene_nocs_pose(ob) for ob in objs]
    return [(s, workload.segment_poses_host(objs, gripper, nocs, s)) for s in segs], objs, gripper, nocs, cats, n_total


def run_batch(batch, objs, gripper, nocs, cats):
    """-> codes (E,), nudge (E,), poses (E,4,4) of the whole batch in global evaluation order under the active variant."""
    codes, nudges, poses = [], [], []
    g = gripper
    bgs = {}
    for seg, P in batch:
        if seg.obj not in bgs:
            bgs[seg.obj] = synth.background_points(objs, seg.obj, g['diameter'])
        sym = transforms.get_symmetry_tfs(cats[seg.obj]) if seg.kind == 'nocs' else [I4]
        nocs_pose = nocs[seg.obj] if seg.kind == 'nocs' else I4
        c, p, n = co.filter_grasp_pose(P, sym, nocs_pose, I4, I4, I4, g['gripper_in_grasp'], 1, 0, int(seg.adjust), g['vertices'], g['faces'],
                                       g['enclosed_vertices'], g['enclosed_faces'], objs[seg.obj]['xyz'], bgs[seg.obj], 0.0005)
        codes.append(c); nudges.append(n); poses.append(p)
    return np.concatenate(codes), np.concatenate(nudges), np.concatenate(poses)


def grazing_pairs(n, seed=0):
    """Triangles placed so that their closest approach to a leaf cube (edge 0.5 mm, at ~0.6 m like the scene) is within +-2e-6 m of
    contact, through a face, an edge or a corner: the cases where the formulations can differ."""
    rng = np.random.default_rng(seed)
    res = np.float32(0.0005)
    keys = rng.integers(-400, 1400, (n, 3)).astype(np.int32); keys[:, 2] = rng.integers(1100, 1500, n)       # z ~ 0.55 .. 0.75 m
    c = ((keys.astype(np.float32) + np.float32(0.5)) * res).astype(np.float64)
    h = 0.5 * float(res)
    kind = rng.integers(0, 3, n)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    s = np.sign(d)
    face = np.zeros((n, 3)); face[np.arange(n), rng.integers(0, 3, n)] = 1.0
    off = np.where(kind[:, None] == 0, face * s, np.where(kind[:, None] == 1, (1 - face) * s, s)) * h             # face centre / edge midpoint / corner
    gap = rng.uniform(-2e-6, 2e-6, n)
    nrm = off / np.maximum(np.linalg.norm(off, axis=1, keepdims=True), 1e-30)
    p0 = c + off + nrm * gap[:, None]                                                                             # nearest point of the triangle
    t1 = np.cross(nrm, rng.normal(size=(n, 3))); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    size = rng.uniform(0.001, 0.02, (n, 1))
    a = p0
    b = p0 + (t1 * rng.uniform(0.2, 1, (n, 1)) + nrm * rng.uniform(0, 0.5, (n, 1))) * size
    e = p0 + (t2 * rng.uniform(0.2, 1, (n, 1)) + nrm * rng.uniform(0, 0.5, (n, 1))) * size
    return keys, a.astype(np.float32), b.astype(np.float32), e.astype(np.float32)


def grazing_decisions(keys, a, b, e, leaf_mode, dh, narrow):
    """Per-pair decision through cr_mesh_voxels_collide (one triangle, one voxel) under a variant."""
    set_variant(leaf_mode, dh, narrow)
    F = np.array([[0, 1, 2]], dtype=np.int32)
    out = np.zeros(len(keys), dtype=bool)
    for i in range(len(keys)):
        out[i] = co.mesh_voxels_collide(np.stack([a[i], b[i], e[i]]), F, I4, keys[i:i + 1], 0.0005)
    set_variant()
    return out


def main(out_path, per_replica=50000, n_grazing=100000):
    t0 = time.time()
    batch, objs, gripper, nocs, cats, n_total = c3_batch(per_replica=per_replica)
    set_variant()
    base = run_batch(batch, objs, gripper, nocs, cats)
    report = {'what': 'sensitivity of the filterGraspPose result to the formulation of the leaf-box / triangle predicate (FCL and octomap absent: '
                      'PARITY UNPINNED); baseline = oracle/collision_ref.c (float64 clipping); sat_f32 = the arithmetic of csrc/collision.hip',
              'workload': f'C3 (BASELINE.json configs[2]): {len(objs)} objects x {len(objs[0]["xyz"])} pts, {n_total} evaluations in the global order of '
                          'catgrasp_amd/workload.py, both call shapes (grasp_sampler.py:345 with 12 symmetries and pose nudging; :216), resolution 0.0005',
              'baseline_code_histogram_0keep_1dir_2ik_3open_4enclosed': np.bincount(base[0], minlength=5).tolist(),
              'baseline_nudge_histogram_-1..4': np.bincount(base[1] + 1, minlength=6).tolist(), 'variants': {}}
    union = np.zeros(n_total, dtype=bool)
    for name, (lm, dh, nar) in VARIANTS.items():
        set_variant(lm, dh, nar)
        c, n, p = run_batch(batch, objs, gripper, nocs, cats)
        set_variant()
        code_flip = c != base[0]
        nudge_flip = (n != base[1]) & ~code_flip
        surv = (c == 0) != (base[0] == 0)
        both_keep = (c == 0) & (base[0] == 0)
        pose_diff = both_keep & (np.abs(p - base[2]).reshape(n_total, -1).max(1) > 0)
        any_change = code_flip | nudge_flip | pose_diff
        if '+' not in name:
            union |= any_change
        report['variants'][name] = {'leaf_mode': lm, 'half_edge_delta_m': dh, 'narrow_phase': NARROW_NAME[nar],
                                    'codes_flipped': int(code_flip.sum()), 'nudge_index_changed': int(nudge_flip.sum()),
                                    'survivor_set_changed': int(surv.sum()), 'kept_pose_changed': int(pose_diff.sum()),
                                    'evaluations_with_any_change': int(any_change.sum()),
                                    'flip_matrix_base_to_variant': {f'{a}->{b}': int(((base[0] == a) & (c == b)).sum())
                                                                    for a in range(5) for b in range(5) if a != b and ((base[0] == a) & (c == b)).any()}}
    report['evaluations_inside_the_fcl_uncertainty_band'] = int(union.sum())
    report['evaluations_total'] = int(n_total)
    report['band_definition'] = 'an evaluation is inside the band if its code, its accepted nudge or its kept pose changes under ANY of: sat_f32, mpr_libccd, fcl_halving, grow_1um, shrink_1um'
    keys, a, b, e = grazing_pairs(n_grazing)
    gbase = grazing_decisions(keys, a, b, e, 0, 0.0, 0)
    gz = {'pairs': int(n_grazing), 'gap_range_m': [-2e-6, 2e-6], 'baseline_overlaps': int(gbase.sum())}
    for name, (lm, dh, nar) in VARIANTS.items():
        gz[name + '_flips'] = int((grazing_decisions(keys, a, b, e, lm, dh, nar) != gbase).sum())
    report['grazing_pairs'] = gz
    report['wall_s'] = round(time.time() - t0, 1)
    report['oracle_threads'] = co.num_threads()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    return report


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r4_collision_sensitivity.json'))
    ap.add_argument('--evaluations', type=int, default=50000)
    ap.add_argument('--grazing', type=int, default=100000)
    a = ap.parse_args()
    main(a.out, a.evaluations, a.grazing)
