"""The hot path over one scene as ONE shardable batch of candidate evaluations -- what bench.py times and what the multi-GPU
path partitions (catgrasp_amd/distributed.py).

Per object the reference's CombinedGraspSampler (dexnet/grasping/grasp_sampler.py:360-370) produces candidates through two
filterGraspPose call shapes, and every survivor is scored by GraspPredicter.predict_batch (run_grasp_simulation.py:296-319):
  * 'nocs' segment  grasp_sampler.py:345 -- canonical grasps x the category's symmetry transforms (nut 12, hnm 2, screw 72;
                    Utils.py:79-94), nocs_pose = the object's NUNOCS 9-D pose, adjust_collision_pose=True;
  * 'cone' segment  grasp_sampler.py:216 -- camera-frame cone-sampler poses, symmetry_tfs=[I], nocs_pose=I.
A candidate = one (pose, symmetry) evaluation.  The batch is the concatenation of all segments in a fixed global order; any
contiguous slice [lo, hi) of it can be evaluated independently of the rest (read-only scene data is replicated), and the result
of a slice does not depend on how the batch was cut: shard union == whole, bit for bit (tests/test_workload_gpu.py).

The planning / slicing arithmetic is plain python (covered on CPU, also under a 2-rank gloo group with the device stages replaced
at the tensor level: tests/test_distributed_gloo.py); `SceneBatch` holds the device state and issues the kernels.
"""
from dataclasses import dataclass

import numpy as np

SYMMETRY_COUNT = {'nut': 12, 'hnm': 2, 'screw': 72}        # Utils.py:79-94


@dataclass(frozen=True)
class Segment:
    replica: int        # weak scaling: which rank's copy of the candidate set (same scene, different candidate seeds)
    obj: int
    kind: str           # 'nocs' | 'cone'
    n_pose: int
    n_sym: int
    start: int          # global index of the segment's first evaluation
    adjust: bool

    @property
    def count(self):
        return self.n_pose * self.n_sym


def plan_segments(n_objects, per_replica, n_sym, replicas=1, nocs_fraction=0.5):
    """Global evaluation order: replica-major, then object, then [nocs segment, cone segment].  `per_replica` evaluations are
    split evenly over the objects (remainder to the first ones); within an object ~nocs_fraction of them come from canonical
    grasps x n_sym symmetries (adjust=True) and the rest from cone poses (adjust=False).  n_sym: the category's symmetry count, or
    one count per object for a mixed-category bin.  -> (segments, n_total)."""
    per_obj_sym = [int(n_sym)] * n_objects if np.ndim(n_sym) == 0 else [int(v) for v in n_sym]
    assert len(per_obj_sym) == n_objects and all(v >= 1 for v in per_obj_sym)
    segs, start = [], 0
    for r in range(replicas):
        for k in range(n_objects):
            n_sym = per_obj_sym[k]
            per = per_replica // n_objects + (1 if k < per_replica % n_objects else 0)
            n_can = int(round(per * nocs_fraction / n_sym))
            n_cone = per - n_can * n_sym
            if n_cone < 0:
                n_can, n_cone = per // n_sym, per - (per // n_sym) * n_sym
            for kind, n_pose, ns, adj in (('nocs', n_can, n_sym, True), ('cone', n_cone, 1, False)):
                if n_pose > 0:
                    segs.append(Segment(r, k, kind, n_pose, ns, start, adj))
                    start += n_pose * ns
    return segs, start


def intersect(segs, lo, hi):
    """Segments touched by the global slice [lo, hi) with the local evaluation range of each: [(segment, a, b)]."""
    out = []
    for s in segs:
        a, b = max(lo, s.start) - s.start, min(hi, s.start + s.count) - s.start
        if a < b:
            out.append((s, a, b))
    return out


def split_eval_range(n_sym, a, b):
    """Evaluations [a, b) of an (n_pose x n_sym) block (e = i*n_sym + j) as at most three full rectangles
    (i0, i1, j0, j1) in evaluation order -- what one filterGraspPose call (poses [i0,i1) x symmetries [j0,j1)) can express:
    a partial head group, the whole groups in the middle, a partial tail group."""
    out = []
    if a >= b:
        return out
    i0, j0 = divmod(a, n_sym)
    i1, j1 = divmod(b, n_sym)
    if i0 == i1:
        return [(i0, i0 + 1, j0, j1)]
    if j0:
        out.append((i0, i0 + 1, j0, n_sym)); i0 += 1
    if i1 > i0:
        out.append((i0, i1, 0, n_sym))
    if j1:
        out.append((i1, i1 + 1, 0, j1))
    return out


def segment_poses_host(objs, gripper, nocs_poses, seg):
    """Candidate poses (n_pose,4,4) float64 of a segment: a deterministic function of (replica, object, kind), so any rank -- and the
    CPU-side studies under oracle/ -- can build any segment without a device.  'nocs' poses live in the object's NUNOCS frame
    (grasp_sampler.py:337-339)."""
    from . import synth
    rng = np.random.default_rng([1000 + seg.replica, seg.obj, 0 if seg.kind == 'nocs' else 1])
    P = synth.make_candidates(objs[seg.obj], seg.n_pose, rng, gripper['hand_depth'], gripper['init_bite'])
    if seg.kind == 'nocs':
        P = np.linalg.inv(nocs_poses[seg.obj]) @ P
    return P


def segment_poses_many(objs, gripper, nocs_poses, segs, workers=None, min_poses=40000):
    """segment_poses_host for many segments -> [poses].  The candidate generator is a seeded python loop per pose (rejection sampling
    of the approach direction: the values are defined by its sequential stream), 0.1 ms per pose: 23 s for the 500,000-candidate bin
    on one core.  Segments are independent (own generator each), so big batches are generated by worker PROCESSES (plain
    interpreters that import numpy only: `python -m catgrasp_amd.workload <in> <out>`), the segments dealt out by size; the result is
    the serial one bit for bit.  Any failure of the worker route falls back to the serial loop."""
    import os
    import pickle
    import shutil
    import subprocess
    import sys
    import tempfile
    total = sum(s.n_pose for s in segs)
    w = workers if workers is not None else min(len(segs), 16, max(1, (os.cpu_count() or 2) // 2))
    serial = lambda: [segment_poses_host(objs, gripper, nocs_poses, s) for s in segs]
    if total < min_poses or w < 2:
        return serial()
    bins = [[] for _ in range(w)]
    load = [0] * w
    for i in sorted(range(len(segs)), key=lambda i: -segs[i].n_pose):        # largest first onto the lightest worker
        k = load.index(min(load)); bins[k].append(i); load[k] += segs[i].n_pose
    tmp = tempfile.mkdtemp(prefix='cg_poses_')
    try:
        slim = [{'xyz': o['xyz'], 'normal': o['normal']} for o in objs]
        g = {'hand_depth': gripper['hand_depth'], 'init_bite': gripper['init_bite']}
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1')
        procs = []
        for k, idx in enumerate(bins):
            if not idx:
                continue
            fin, fout = os.path.join(tmp, f'in{k}.pkl'), os.path.join(tmp, f'out{k}.npz')
            with open(fin, 'wb') as f:
                pickle.dump((slim, g, [np.asarray(p) for p in nocs_poses], [segs[i] for i in idx]), f)
            procs.append((idx, fout, subprocess.Popen([sys.executable, '-m', 'catgrasp_amd.workload', fin, fout], env=env, cwd=root,
                                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        out = [None] * len(segs)
        for idx, fout, pr in procs:
            if pr.wait(timeout=600) != 0:
                raise RuntimeError('pose worker failed')
            with np.load(fout) as z:
                for j, i in enumerate(idx):
                    out[i] = z[f'p{j}']
        return out
    except Exception:
        return serial()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _poses_worker(fin, fout):
    import pickle
    with open(fin, 'rb') as f:
        objs, g, nocs_poses, segs = pickle.load(f)
    np.savez(fout, **{f'p{j}': segment_poses_host(objs, g, nocs_poses, s) for j, s in enumerate(segs)})


def scene_nocs_pose(ob, nocs_scale=0.02):
    """The synthetic 9-D NUNOCS pose of a scene object (object pose x isotropic scale)."""
    return ob['pose'] @ np.diag([nocs_scale, nocs_scale, nocs_scale, 1.0])


def pack_records(p_g, codes):
    """The per-candidate record that crosses xGMI: (p_G, reject code) as two float32 lanes, 8 B/candidate."""
    import torch
    return torch.stack([p_g.float(), codes.float()], dim=1).contiguous()


class SceneBatch:
    """Device state of one synthetic scene + its candidate batch (SURVEY.md §8(d) inputs).  `score_slice(lo, hi)` runs the hot
    path over global evaluations [lo, hi): NUNOCS net over the slice's objects -> filterGraspPose (both call shapes) -> device
    pose inversion + resampling draw + grasp-Q net -> packed (p_G, code) records."""

    # 'multi': the filter calls of a slice as ONE launch sequence (run_filter_many); 'per_segment': one filterGraspPose call per (object,
    # call shape) on the objects' side streams -- the round-5 form, kept for the equality test (tests/test_workload_gpu.py)
    filter_launch = 'multi'

    def __init__(self, device, grasp_predicter, nunocs_predicter, kind='nut', n_objects=8, pts_per_object=2500, per_replica=50000,
                 replicas=1, scene_seed=0, nocs_scale=0.02, materialize=None, gripper_subdivisions=0):
        # materialize: (lo, hi) global evaluation range whose candidate poses are generated up front (default: all)
        # kind: one category ('nut' | 'hnm' | 'screw'), or a mixed bin of synth.MIXED_BINS ('bin' = nut + hnm + screw, BASELINE.json
        # configs[4]); for a mixed bin the two predicters are dicts {category: predicter} -- the reference keeps one GraspPredicter /
        # NunocsPredicter per class (run_grasp_simulation.py:701-702), each with its own weights.
        import torch
        from . import my_cpp, synth, transforms
        self.device, self.kind = device, kind
        self.objs = synth.make_scene(n_objects, pts_per_object, seed=scene_seed, kind=kind)       # same scene on every rank
        # gripper_subdivisions: 0 = the 36 / 48-triangle box gripper, 4 = the same surfaces as 9,216 / 12,288 triangles (bench.py)
        self.gripper = synth.make_gripper(subdivisions=gripper_subdivisions)
        self.cats = [ob['kind'] for ob in self.objs]                                               # category of every object
        self.gps = grasp_predicter if isinstance(grasp_predicter, dict) else {c: grasp_predicter for c in set(self.cats)}
        self.npreds = nunocs_predicter if isinstance(nunocs_predicter, dict) else {c: nunocs_predicter for c in set(self.cats)}
        missing = set(self.cats) - set(self.gps) | set(self.cats) - set(self.npreds)
        if missing:
            raise KeyError(f'no predicter for categories {sorted(missing)}')
        self.gp, self.npred = self.gps[self.cats[0]], self.npreds[self.cats[0]]
        if len({g.cfg['n_pts'] for g in self.gps.values()}) != 1:
            raise ValueError('the grasp predicters of one bin must resample to the same n_pts')
        self.n_sym = max(SYMMETRY_COUNT[c] for c in self.cats)
        self.segs, self.n_total = plan_segments(n_objects, per_replica, [SYMMETRY_COUNT[c] for c in self.cats], replicas)
        self.syms = {c: torch.from_numpy(np.stack(transforms.get_symmetry_tfs(c)).astype(np.float32).reshape(-1, 16)).to(device)
                     for c in set(self.cats)}
        self.eye = torch.eye(4, device=device).reshape(1, 16).contiguous()
        self.clouds, self.offsets, self.scenes, self.nocs_pose = [], [], [], []
        off = 0
        for k, ob in enumerate(self.objs):
            dc = transforms.DeviceCloud(ob['xyz'], ob['normal'], device)
            self.clouds.append(dc); self.offsets.append(off); off += dc.n
            bg = synth.background_points(self.objs, k, self.gripper['diameter'])
            g = self.gripper
            self.scenes.append(my_cpp.GripperScene(g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg,
                                                   0.0005, device))
            self.nocs_pose.append(scene_nocs_pose(ob, nocs_scale))
        self.cloud_xyz = torch.cat([c.xyz for c in self.clouds]).contiguous()
        self.cloud_normal = torch.cat([c.normal for c in self.clouds]).contiguous()
        gen = torch.Generator(device=device); gen.manual_seed(1234)
        n_n = self.npred.cfg['n_pts'] if self.npred is not None else 8192
        self.nunocs_ids = torch.stack([transforms.draw_ids_device(c.n, n_n, 1, device, gen, base=o)[0]
                                       for c, o in zip(self.clouds, self.offsets)]).contiguous()
        self._poses = {}
        self._plans = {}
        self._streams = [torch.cuda.Stream(device=device) for _ in range(n_objects)]
        self.draw_seed = 0x5eed
        lo, hi = (0, self.n_total) if materialize is None else materialize        # a rank only builds the segments it will evaluate
        mine = [s for s, _, _ in intersect(self.segs, lo, hi)]
        for s, P in zip(mine, segment_poses_many(self.objs, self.gripper, self.nocs_pose, mine)):
            self._poses[(s.replica, s.obj, s.kind)] = torch.from_numpy(P.astype(np.float32).reshape(-1, 16)).to(self.device)

    # ---- candidate poses of a segment: a deterministic function of (replica, object, kind), so any rank can build any segment
    def segment_poses(self, seg):
        import torch
        key = (seg.replica, seg.obj, seg.kind)
        if key not in self._poses:
            P = segment_poses_host(self.objs, self.gripper, self.nocs_pose, seg)
            self._poses[key] = torch.from_numpy(P.astype(np.float32).reshape(-1, 16)).to(self.device)
        return self._poses[key]

    def host_poses(self, seg):
        return self.segment_poses(seg).cpu().numpy().astype(np.float64).reshape(-1, 4, 4)

    # ---- device stages (replaced at the tensor level by the CPU control-flow test)
    def run_nunocs(self, obj_ids):
        """NUNOCS net over the given objects, one batched forward per category present (each category has its own weights)."""
        import torch
        out = {}
        for cat in dict.fromkeys(self.cats[k] for k in obj_ids):
            ks = [k for k in obj_ids if self.cats[k] == cat]
            ids = self.nunocs_ids[torch.as_tensor(ks, device=self.device)]
            out[cat] = (ks, self.npreds[cat].nocs_on_device(self.cloud_xyz, self.cloud_normal, ids))
        return out

    def run_filter(self, seg, i0, i1, j0, j1):
        """-> codes (E) int8, grasp_in_cam (E,16) f32 (rejected evaluations keep their composed pose) for poses [i0,i1) x syms [j0,j1)."""
        from . import my_cpp
        I4 = np.eye(4, dtype=np.float32)
        g = self.gripper
        if seg.kind == 'nocs':
            sym, nocs = self.syms[self.cats[seg.obj]][j0:j1], self.nocs_pose[seg.obj]
        else:
            sym, nocs = self.eye, I4
        codes, poses, _ = my_cpp.filter_on_device(self.scenes[seg.obj], self.segment_poses(seg)[i0:i1], sym, nocs, I4, I4, I4,
                                                  g['gripper_in_grasp'], True, False, seg.adjust, keep_rejected_pose=True)
        return codes, poses.view(-1, 16)

    def run_filter_many(self, key, rects):
        """Device stage: the filter over the rectangles [(segment, i0, i1, j0, j1)] as ONE launch sequence (my_cpp.FilterPlan /
        cg_filter_grasp_pose_multi) -> codes (E) int8, grasp_in_cam (E,16) f32 in rectangle order.  The segment table of `key` (a slice)
        is built and uploaded once and reused by every later step over the same slice."""
        from . import my_cpp
        plan = self._plans.get(key)
        if plan is None:
            I4 = np.eye(4, dtype=np.float32)
            rows = []
            for s, i0, i1, j0, j1 in rects:
                if s.kind == 'nocs':
                    sym, nocs = self.syms[self.cats[s.obj]][j0:j1], self.nocs_pose[s.obj]
                else:
                    sym, nocs = self.eye, I4
                rows.append((self.scenes[s.obj], self.segment_poses(s)[i0:i1], sym, nocs, I4, s.adjust))
            plan = self._plans[key] = my_cpp.FilterPlan(rows)
        codes, poses, _ = plan.run(self.gripper['gripper_in_grasp'], True, keep_rejected_pose=True)
        return codes, poses.view(-1, 16)

    def run_prep(self, obj, poses, row_offset, pinv_out, ids_out):
        """grasp_in_cam (E,16) of one object -> rows of the scoring inputs: device pose inversion (re-expressed for the object's
        centred cloud) and the per-candidate resampling draw of GraspDataset.transform, keyed by the GLOBAL evaluation index."""
        from . import transforms
        dc = self.clouds[obj]
        pinv_out.copy_(transforms.pose_inverse_rows_device(poses, dc.center))
        transforms.draw_ids_device(dc.n, self.gp.cfg['n_pts'], poses.shape[0], self.device, seed=self.draw_seed,
                                   base=self.offsets[obj], row_offset=row_offset, out=ids_out)

    def run_net(self, ids, pinv, cat=None):
        """-> p_G (E): input transform + PointNetCls + softmax + p_G for every candidate of the rows in one batch, under the
        weights of category `cat` (None: the bin's only category)."""
        gp = self.gp if cat is None else self.gps[cat]
        return gp.score_on_device(self.cloud_xyz, self.cloud_normal, ids, pinv)[3]

    def alloc(self, n):
        import torch
        return (torch.empty((n, 12), dtype=torch.float32, device=self.device),
                torch.empty((n, self.gp.cfg['n_pts']), dtype=torch.int32, device=self.device))

    # ---- the step
    def score_slice(self, lo, hi):
        import torch
        parts = intersect(self.segs, lo, hi)
        if not parts:
            return torch.empty((0, 2), dtype=torch.float32, device=self.device)
        self.run_nunocs(sorted({s.obj for s, _, _ in parts}))
        # filter: the per-segment kernels are small (a few thousand wavefronts), so objects run concurrently on side streams
        import contextlib
        on_gpu = self.device.type == 'cuda'
        if on_gpu:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event(); fork.record(main)
        if self.filter_launch == 'multi':
            codes, poses = self.run_filter_many((lo, hi), [(s, *r) for s, a, b in parts for r in split_eval_range(s.n_sym, a, b)])
        else:
            codes, poses = [], []
            for s, a, b in parts:
                st = self._streams[s.obj] if on_gpu else None
                with (torch.cuda.stream(st) if on_gpu else contextlib.nullcontext()):
                    if on_gpu:
                        st.wait_event(fork)
                    for i0, i1, j0, j1 in split_eval_range(s.n_sym, a, b):
                        c, p = self.run_filter(s, i0, i1, j0, j1)
                        codes.append(c); poses.append(p)
            if on_gpu:
                for st in {self._streams[s.obj] for s, _, _ in parts}:
                    main.wait_stream(st)
            codes = torch.cat(codes)
            poses = torch.cat(poses)
        # scoring inputs: one prep per run of consecutive evaluations of the same object (the pose inverse is per object cloud)
        runs = []
        for s, a, b in parts:
            if runs and runs[-1][0] == s.obj and runs[-1][1] + runs[-1][2] == s.start + a:
                runs[-1][2] += b - a
            else:
                runs.append([s.obj, s.start + a, b - a])
        pinv, ids = self.alloc(hi - lo)
        pos = 0
        if on_gpu:      # the resampling draw is a ~0.3 ms dependent chain per launch whatever its size: the objects' preps run side by side
            fork = torch.cuda.Event(); fork.record(main)
        for obj, g0, n in runs:
            st = self._streams[obj] if on_gpu else None
            with (torch.cuda.stream(st) if on_gpu else contextlib.nullcontext()):
                if on_gpu:
                    st.wait_event(fork)
                self.run_prep(obj, poses[pos:pos + n], g0, pinv[pos:pos + n], ids[pos:pos + n])
            pos += n
        if on_gpu:
            for st in {self._streams[obj] for obj, _, _ in runs}:
                main.wait_stream(st)
        # scoring: one batch per run of consecutive rows of the same category (a single-category bin: the whole slice at once)
        spans = []
        pos = 0
        for obj, _, n in runs:
            if spans and spans[-1][0] == self.cats[obj]:
                spans[-1][2] += n
            else:
                spans.append([self.cats[obj], pos, n])
            pos += n
        if len(spans) == 1:
            p_g = self.run_net(ids, pinv, spans[0][0])
        else:
            p_g = torch.cat([self.run_net(ids[a:a + n], pinv[a:a + n], cat) for cat, a, n in spans])
        return pack_records(p_g, codes)


def build_flat_workload(device, G, seed, n_objects=8, pts_per_object=2500, kind='nut'):
    """A pre-expanded candidate batch (camera-frame poses, host-inverted pose rows and resample ids already on the device): the
    inputs of score_on_device / filter_on_device without the step logic.  Used by the full-size property tests."""
    import torch
    from . import my_cpp, synth, transforms
    objs = synth.make_scene(n_objects, pts_per_object, seed=0, kind=kind)
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
        ids.append(transforms.draw_ids_device(dc.n, 2048, per[k], device, gen, base=off))
        off += dc.n
    return {'objs': objs, 'gripper': gripper, 'per': per, 'scenes': scenes, 'poses_dev': poses_dev,
            'cloud_xyz': torch.cat([c.xyz for c in clouds]).contiguous(),
            'cloud_normal': torch.cat([c.normal for c in clouds]).contiguous(),
            'ids': torch.cat(ids).contiguous(),
            'pose_inv': torch.from_numpy(np.concatenate(pose_rows)).to(device),
            'nunocs_ids': torch.stack([transforms.draw_ids_device(c.n, 8192, 1, device, gen, base=o)[0]
                                       for c, o in zip(clouds, offsets)]).contiguous(),
            'G': G}


if __name__ == '__main__':          # a pose worker of segment_poses_many
    import sys
    _poses_worker(sys.argv[1], sys.argv[2])
