"""One object's candidate-grasp evaluation, device resident end to end: the MI355X counterpart of
`compute_candidate_grasp_one_ob` + the scoring loop of `compute_candidate_grasp`
(run_grasp_simulation.py:112-183, :296-329), assembled from this package's drop-in pieces:

    background occupancy   my_cpp.makeOccupancyGridFromCloudScan          run_grasp_simulation.py:131-139
    NUNOCS + 9-D pose      NunocsPredicter.predict (net + device RANSAC)   :146
    candidates             grasp_sampler.cone_grasp_poses [+ canonical grasps x symmetries]   :176 -> grasp_sampler.py
    collision/approach     my_cpp.filter_on_device                         grasp_sampler.py:216,345
    affordance P(T|G)      affordance.compute_grasp_affordance             :181
    grasp quality P(G)     GraspPredicter.score_on_device                  :310-313
    ranking                P(T,G) = P(T|G) P(G), descending                :314-329

Host work is limited to small bookkeeping (index draws, 4x4 inverses); every per-point / per-candidate loop runs in a
HIP kernel.  Used by examples/run_scene.py and tests/test_pipeline_gpu.py.
"""
import time

import numpy as np
import torch

from . import affordance as aff_mod
from . import grasp_sampler, my_cpp, transforms


def _background_points(scene_pts, ob_pts, gripper_diameter, device):
    """run_grasp_simulation.py:127-135: scene points within gripper_diameter/2 of the object, minus the object itself
    (cloudA_minus_cloudB thres 5 mm), voxel-downsampled to 1 mm (one representative per voxel)."""
    nn = aff_mod.nearest_neighbor(scene_pts, ob_pts, device).cpu().numpy()
    d = np.linalg.norm(np.asarray(scene_pts) - np.asarray(ob_pts)[nn], axis=1)
    bg = np.asarray(scene_pts)[(d <= gripper_diameter / 2) & (d > 0.005)]
    if len(bg) == 0:
        return bg.reshape(0, 3)
    keys = np.floor(bg / 0.001).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)
    return bg[np.sort(first)]


def canonical_fields(canonical):
    """The canonical model of a category in the form evaluate_object uses -- dict(cloud, normals, affordance, grasps (m,4,4)).  Accepts
    that form, or the reference's own `{class}_canonical.pkl` dict (grasp_sampler.load_canonical: 'canonical_cloud',
    'canonical_normals', 'canonical_affordance', 'canonical_grasps' = grasp objects; run_grasp_simulation.py:706-707)."""
    if canonical is None or 'canonical_cloud' not in canonical:
        return canonical
    grasps = list(canonical['canonical_grasps'])
    return {'cloud': np.asarray(canonical['canonical_cloud'], dtype=np.float64), 'normals': np.asarray(canonical['canonical_normals'], dtype=np.float64),
            'affordance': np.asarray(canonical['canonical_affordance'], dtype=np.float64).reshape(-1),
            'grasps': np.stack([g.get_grasp_pose_matrix() for g in grasps]) if grasps else np.zeros((0, 4, 4))}


def evaluate_object(ob_pts, ob_normals, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, canonical=None, symmetry_tfs=None,
                    n_surface_samples=50, sphere_pts=None, approach_step=0.004, resolution=0.0005, cam_in_world=None, timings=None, ik=None,
                    rng=None, nocs_pose_override=None, nunocs_predrawn=None, on_scoring_draws=None):
    """Returns dict(poses (n,4,4) f32, p_G, p_T_given_G, p_T_G, order) for the surviving candidates, best first.
    `gripper`: dict with vertices/faces/enclosed_vertices/enclosed_faces/gripper_in_grasp/hand_depth/init_bite/diameter and
    finger_vertices (list of 2 arrays), grip_dirs.  `canonical`: optional dict(cloud, normals, affordance, grasps (m,4,4))
    in the canonical (NUNOCS-scaled) frame, or the reference's `{class}_canonical.pkl` dict as grasp_sampler.load_canonical returns it
    (canonical_fields); without it P(T|G) = 1 and only cone-sampled candidates are produced.
    `ik`: optional dict(ee_in_grasp 4x4, upper[7], lower[7]) -> filter_ik=True with the device iiwa14 solver (cam_in_world must
    then be the camera pose in the robot base frame, common.cpp:214-226).
    `rng`: the per-candidate resampling draw of the grasp-Q stage -- 'numpy' (numpy's global stream, as predict_batch's default),
    'device' (counter-based device draw); default: the predicter's own setting.  `nocs_pose_override`: use this 4x4 for the canonical
    branch instead of the RANSAC result (NunocsPredicter.predict still runs and is timed): random-init weights cannot recover a pose.
    `timings` additionally receives the split of the NUNOCS stage that NunocsPredicter.predict records (net / id draw / RANSAC).
    `nunocs_predrawn` / `on_scoring_draws`: the two ends of evaluate_objects' draw-ahead (NunocsPredicter.draw_ahead's result for this
    object; a callback(state, n_valid, n_pts, n_rows) fired when this object's scoring pass starts drawing from numpy's stream).
    = prepare_object (every stage before the scoring pass) + score_object, the two halves evaluate_objects overlaps across objects."""
    prep = prepare_object(ob_pts, ob_normals, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, canonical=canonical, symmetry_tfs=symmetry_tfs,
                          n_surface_samples=n_surface_samples, sphere_pts=sphere_pts, approach_step=approach_step, resolution=resolution,
                          cam_in_world=cam_in_world, timings=timings, ik=ik, nocs_pose_override=nocs_pose_override, nunocs_predrawn=nunocs_predrawn)
    return score_object(prep, grasp_predicter, rng=rng, timings=timings, on_scoring_draws=on_scoring_draws)


def _lap(timings, name, t0):
    if timings is not None:
        torch.cuda.current_stream().synchronize()          # this thread's stream only: another object's stages may be running beside it
        timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0)


def prepare_object(ob_pts, ob_normals, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, canonical=None, symmetry_tfs=None,
                   n_surface_samples=50, sphere_pts=None, approach_step=0.004, resolution=0.0005, cam_in_world=None, timings=None, ik=None,
                   nocs_pose_override=None, nunocs_predrawn=None, np_state=None):
    """Every stage of evaluate_object BEFORE the grasp-Q scoring pass: background occupancy, NUNOCS + 9-D pose, candidate generation,
    collision / approach filter, affordance.  -> the dict score_object consumes.
    np_state None: the stages draw from numpy's GLOBAL generator, in the reference's order (NUNOCS resampling row, 2 x 10,000 RANSAC
    hypotheses, the surface-sample choice).  np_state = a generator state (or a callable returning one, evaluated on the draw-ahead
    thread): the same draws are replayed from THAT state without touching the global generator -- the form evaluate_objects uses to
    run object k+1's stages while object k is still being scored; prep['np_state'] is then where the stream stands when this object's
    scoring pass begins.  Needs a predicter whose draws can be made ahead (NunocsPredicter._predraw)."""
    dev = grasp_predicter.device
    t = time.perf_counter
    canonical = canonical_fields(canonical)
    lap = lambda name, t0: _lap(timings, name, t0)
    I4 = np.eye(4, dtype=np.float32)
    cam_in_world = I4 if cam_in_world is None else cam_in_world
    explicit = np_state is not None
    data = {'cloud_xyz': ob_pts, 'cloud_normal': ob_normals}
    if explicit:        # the sequential host replay of this object's draws starts NOW, under the device stages below
        from . import predicter as pred_mod
        n_valid = int(transforms.valid_mask(np.asarray(ob_pts, dtype=np.float64)).sum())
        nunocs_predrawn = nunocs_predicter.draw_ahead(n_valid, np_state, pred_mod.draw_ahead_worker())
        if nunocs_predrawn is None:
            raise ValueError('prepare_object(np_state=...) needs a NunocsPredicter that draws from numpy\'s stream ahead of time')
    # --- background occupancy (occluded space behind the visible neighbours counts as occupied) ---
    t0 = t()
    bg = _background_points(scene_pts, ob_pts, gripper['diameter'], dev)
    occ = my_cpp.makeOccupancyGridFromCloudScan(bg, K, 0.001, return_tensor=True) if len(bg) else torch.zeros((0, 3), device=dev)
    lap('occupancy', t0)
    # --- NUNOCS + 9-D pose ---
    t0 = t()
    if nunocs_predrawn is not None:
        nunocs_predrawn = nunocs_predrawn.result() if hasattr(nunocs_predrawn, 'result') else nunocs_predrawn
        nocs_cloud, nocs_pose = nunocs_predicter.predict(data, predrawn=nunocs_predrawn, **({'explicit_stream': True} if explicit else {}))
    else:
        nocs_cloud, nocs_pose = nunocs_predicter.predict(data)
    lap('nunocs+ransac', t0)
    if timings is not None:
        for k, v in getattr(nunocs_predicter, 'timings', {}).items():
            timings[k] = timings.get(k, 0.0) + v
    if nocs_pose_override is not None:
        nocs_pose = np.asarray(nocs_pose_override, dtype=np.float64)
    # --- candidates ---
    t0 = t()
    if explicit:
        rs = np.random.RandomState()
        rs.set_state(nunocs_predrawn['end_state'])
        rng_ids = rs.choice(len(ob_pts), size=min(n_surface_samples, len(ob_pts)), replace=False)
        state_before_scoring = rs.get_state()
    else:
        rng_ids = np.random.choice(len(ob_pts), size=min(n_surface_samples, len(ob_pts)), replace=False)
        state_before_scoring = None
    if sphere_pts is None:
        g = np.random.default_rng(0).normal(size=(30, 3)); g[:, 0] = np.abs(g[:, 0]) + 1.0       # directions within a cone about +x
        sphere_pts = g / np.linalg.norm(g, axis=1, keepdims=True)
    cone = grasp_sampler.cone_grasp_poses(ob_pts, ob_normals, rng_ids, sphere_pts, r_ball=0.003, hand_depth=gripper['hand_depth'],
                                          init_bite=gripper['init_bite'], approach_step=approach_step, center_ob_between_gripper=True,
                                          device=dev, return_tensor=True)
    lap('candidate generation', t0)
    # --- collision / approach filter ---
    t0 = t()
    scene = my_cpp.GripperScene(gripper['vertices'], gripper['faces'], gripper['enclosed_vertices'], gripper['enclosed_faces'], ob_pts, occ,
                                resolution, dev)
    sym1 = torch.eye(4, device=dev).reshape(1, 16)
    ee_in_grasp = I4 if ik is None else np.asarray(ik['ee_in_grasp'])
    ik_kw = {} if ik is None else dict(upper=list(ik['upper']), lower=list(ik['lower']))
    with_canonical = canonical is not None and nocs_pose is not None and len(canonical.get('grasps', []))
    sym = symmetry_tfs if symmetry_tfs is not None else [np.eye(4)]
    cone16 = cone.float().reshape(-1, 16).contiguous()
    if ik is None and cone16.shape[0] > 0:
        # both call shapes of the object (grasp_sampler.py:216: cone poses, symmetry [I]; :345: canonical grasps x symmetries under the
        # NUNOCS pose) as ONE launch sequence: each is a few thousand evaluations, a fraction of what fills the chip
        rows = [(scene, cone16, sym1, I4, I4, True)]
        if with_canonical:
            f16 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64), dtype=np.float32).reshape(-1, 16)).to(dev)
            rows.append((scene, f16(canonical['grasps']), f16(sym), nocs_pose, I4, True))
        plan = my_cpp.FilterPlan(rows)
        codes, poses, _ = plan.run(gripper['gripper_in_grasp'], True)
        surv = [poses[codes == 0]]
        n_evaluated = plan.E
    else:               # with the IK stage (its pre-pass computes ee_in_base per call) the calls stay separate
        codes, poses, _ = my_cpp.filter_on_device(scene, cone16, sym1, I4, I4, cam_in_world, ee_in_grasp,
                                                  gripper['gripper_in_grasp'], True, ik is not None, True, **ik_kw)
        keep = codes == 0
        surv = [poses[keep]]
        n_evaluated = int(codes.numel())
        if with_canonical:
            c2, p2, _ = my_cpp.filter_on_device(scene, np.asarray(canonical['grasps']), np.asarray(sym), nocs_pose, I4, cam_in_world, ee_in_grasp,
                                                gripper['gripper_in_grasp'], True, ik is not None, True, **ik_kw)
            surv.append(p2[c2 == 0]); n_evaluated += int(c2.numel())
    surv = torch.cat(surv).contiguous()
    lap('filterGraspPose', t0)
    n = surv.shape[0]
    prep = {'n_evaluated': n_evaluated, 'nocs_pose': nocs_pose, 'n': n, 'ob_pts': ob_pts, 'ob_normals': ob_normals, 'np_state': state_before_scoring,
            'surv_np': surv.cpu().numpy().astype(np.float64), 'p_t_g': None}
    if n == 0:
        return prep
    # --- affordance ---
    t0 = t()
    if canonical is not None and nocs_pose is not None:
        full = np.asarray(canonical['cloud']) @ nocs_pose[:3, :3].T + nocs_pose[:3, 3]
        nrm = np.asarray(canonical['normals']) @ nocs_pose[:3, :3].T
        sel = np.arange(0, len(full), max(1, len(full) // 2000))
        model = aff_mod.AffordanceModel(full[sel], nrm[sel], full, canonical['affordance'], device=dev)
        prep['p_t_g'] = aff_mod.compute_grasp_affordance(model, prep['surv_np'], gripper['gripper_in_grasp'], gripper['finger_vertices'], gripper['grip_dirs'])
    else:
        prep['p_t_g'] = np.ones(n)
    lap('affordance', t0)
    # --- the scoring pass's device inputs that do not depend on its resampling draw: the object's cloud and the survivors' inverse poses
    t0 = t()
    prep['cloud'] = transforms.DeviceCloud(ob_pts, ob_normals, dev)
    prep['pinv'] = torch.from_numpy(transforms.pose_inverse_rows(prep['surv_np'], prep['cloud'].center)).to(dev)
    lap('scoring inputs (cloud upload, pose inverses)', t0)
    return prep


def score_object(prep, grasp_predicter, rng=None, timings=None, on_scoring_draws=None):
    """The grasp-Q scoring pass over prepare_object's survivors + the ranking by P(T,G) (run_grasp_simulation.py:296-329).  A prep made
    from an explicit generator state first moves numpy's GLOBAL generator to where that object's scoring draws begin -- the position
    the serial loop reaches there -- so the global stream advances exactly as in the reference's loop."""
    dev = grasp_predicter.device
    if prep['np_state'] is not None:
        np.random.set_state(prep['np_state'])
    n, surv_np, p_t_g = prep['n'], prep['surv_np'], prep['p_t_g']
    out = {'n_evaluated': prep['n_evaluated'], 'nocs_pose': prep['nocs_pose']}
    if n == 0:
        out.update(poses=np.zeros((0, 4, 4), np.float32), p_G=np.zeros(0), p_T_given_G=np.zeros(0), p_T_G=np.zeros(0))
        return out
    t0 = time.perf_counter()
    cloud, pinv = prep['cloud'], prep['pinv']
    rng = rng or getattr(grasp_predicter, 'rng', 'device')
    if rng == 'numpy':         # the reference's stream (dataset_grasp.py:72-73), replayed one chunk ahead of the device
        if on_scoring_draws is not None:     # numpy's generator stands at the first of this pass's n resampling draws
            on_scoring_draws(np.random.get_state(), cloud.n, grasp_predicter.cfg['n_pts'], n)
        ids = grasp_predicter._numpy_id_chunks(cloud.n, grasp_predicter.cfg['n_pts'], n)
    else:
        ids = transforms.draw_ids_device(cloud.n, grasp_predicter.cfg['n_pts'], n, dev)
    try:
        _, _, _, p_g = grasp_predicter.score_on_device(cloud.xyz, cloud.normal, ids, pinv)
        p_g = p_g.cpu().numpy().astype(np.float64)
    finally:
        if hasattr(ids, 'close'):
            ids.close()
    _lap(timings, 'grasp-Q scoring', t0)
    valid = np.isfinite(p_t_g)                    # the reference drops grasps without a finger contact (:68-70)
    p_tg = np.where(valid, p_t_g, 0.0) * p_g
    order = np.argsort(-p_tg, kind='stable')
    order = order[valid[order]]
    out.update(poses=surv_np[order].astype(np.float32), p_G=p_g[order], p_T_given_G=p_t_g[order], p_T_G=p_tg[order])
    return out


_STAGE_POOL, _STAGE_STREAMS = [], {}


def _stage_worker():
    """The one thread that runs the NEXT object's pre-scoring stages (evaluate_objects, overlap='stages')."""
    if not _STAGE_POOL:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix='catgrasp-object-stages'))
    return _STAGE_POOL[0]


def _stage_stream(device):
    """The stages thread's stream.  (Scoring on a high-priority stream beside it was measured: 225.6 / 225.8 against 230.4 / 223.7 ms per
    object without, alternating runs of one job -- inside the run-to-run spread, not kept.)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _STAGE_STREAMS:
        _STAGE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _STAGE_STREAMS[key]


def evaluate_objects(objects, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, draw_ahead=True, timings=None, overlap=None, **kw):
    """evaluate_object over the segmented objects of a scene, in order (the loop of compute_candidate_grasp,
    run_grasp_simulation.py:188-329) -- same results, same numpy generator state afterwards as calling evaluate_object per object.

    objects: [dict(ob_pts, ob_normals [, canonical, symmetry_tfs, nocs_pose_override])].  timings: optional list, one dict per object.
    overlap (default: 'stages' when draw_ahead and the predicter's draws can be made ahead, else None):
      'stages' -- object k+1's WHOLE pre-scoring half (prepare_object: occupancy ray cast, NUNOCS network + decode, the 2 x 10,000
          hypothesis draws and the RANSAC kernels, cone sampler, collision filter, affordance) runs on a second thread and a second HIP
          stream while the device scores object k (195 of an object's ~240 ms at the C3 sizes).  What makes this legal under the
          reference's ONE global numpy stream: every draw of the pre-scoring half is replayed from an EXPLICIT generator state -- the
          state the serial loop would have reached there, known as soon as object k's survivor count is (its scoring pass draws exactly
          one resampling row per survivor; transforms.advance_choice_rows walks a copy of the generator over them) -- and the global
          generator is moved to the matching position before each scoring pass (score_object).  Object 0's draws start at entry, under
          its own occupancy and NUNOCS stages.  The stages thread runs up to three objects ahead of the scoring loop (an object with few
          survivors scores faster than its successor prepares; the lead absorbs that).
      'draws'  -- round 5's form: only the next object's NUNOCS-stage draws (~80 ms of sequential host work: numpy's Fisher-Yates
          rejection walk) are made ahead, on a second thread; NunocsPredicter.predict takes them only if numpy's generator really
          stands where they started.
      None     -- the serial loop."""
    from . import engine
    from . import predicter as pred_mod
    rng = kw.get('rng') or getattr(grasp_predicter, 'rng', 'device')
    can_predraw = bool(getattr(nunocs_predicter, '_predraw', False))
    if overlap is None:
        overlap = 'stages' if (draw_ahead and can_predraw and kw.get('ik') is None) else None
    if overlap == 'stages' and not can_predraw:
        overlap = None
    if overlap == 'draws' and not (rng == 'numpy' and can_predraw):
        overlap = None
    if overlap == 'stages':
        return _evaluate_objects_overlapped(objects, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, rng, timings, kw)
    ahead = overlap == 'draws'
    results, pending = [], [None]
    for k, ob in enumerate(objects):
        nxt = objects[k + 1] if k + 1 < len(objects) else None
        hook = None
        if ahead and nxt is not None:
            n_valid_next = int(transforms.valid_mask(np.asarray(nxt['ob_pts'], dtype=np.float64)).sum())

            def hook(state, n_valid, n_pts, n_rows, _nv=n_valid_next):
                pending[0] = nunocs_predicter.draw_ahead(_nv, lambda: transforms.advance_choice_rows(state, n_valid, n_pts, n_rows),
                                                         pred_mod.draw_ahead_worker())
        predrawn, pending[0] = pending[0], None
        tm = {} if timings is not None else None
        try:
            results.append(evaluate_object(ob['ob_pts'], ob['ob_normals'], scene_pts, K, gripper, grasp_predicter, nunocs_predicter,
                                           canonical=ob.get('canonical'), symmetry_tfs=ob.get('symmetry_tfs'),
                                           nocs_pose_override=ob.get('nocs_pose_override'), timings=tm, nunocs_predrawn=predrawn,
                                           on_scoring_draws=hook, **kw))
        except BaseException:
            if pending[0] is not None:
                pending[0].exception()          # let the draw-ahead thread finish before the error travels on
            raise
        if timings is not None:
            timings.append(tm)
    return results


def _evaluate_objects_overlapped(objects, scene_pts, K, gripper, grasp_predicter, nunocs_predicter, rng, timings, kw, depth=3):
    """evaluate_objects(overlap='stages'): see there.  A producer on the stages thread prepares the objects in order, each from the
    generator state the previous one leaves behind (prep k's state before scoring, advanced over the rows its scoring pass will draw:
    both known when prep k is done, so the producer needs nothing from the consumer) and runs up to `depth` objects ahead of the scoring
    loop -- scoring times vary with the survivor counts, a one-object lead would stall the device whenever an object scores faster
    than its successor prepares."""
    import queue
    from . import engine
    dev = grasp_predicter.device
    side, pool, prec = _stage_stream(dev), _stage_worker(), engine.current_precision()
    prep_kw = {k: v for k, v in kw.items() if k != 'rng'}
    n_pts = grasp_predicter.cfg['n_pts']
    main = torch.cuda.current_stream(dev)
    ready = torch.cuda.Event(); ready.record(main)              # the stages may read what the caller's stream produced so far
    tms = [({} if timings is not None else None) for _ in objects]
    q, stop = queue.Queue(maxsize=max(1, depth)), [False]
    state0 = np.random.get_state()

    def produce():
        state = state0
        try:
            with torch.cuda.device(dev), torch.cuda.stream(side), torch.no_grad(), engine.precision(prec):
                side.wait_event(ready)
                for k, ob in enumerate(objects):
                    if stop[0]:
                        return
                    t_start = time.perf_counter()
                    prep = prepare_object(ob['ob_pts'], ob['ob_normals'], scene_pts, K, gripper, grasp_predicter, nunocs_predicter,
                                          canonical=ob.get('canonical'), symmetry_tfs=ob.get('symmetry_tfs'),
                                          nocs_pose_override=ob.get('nocs_pose_override'), timings=tms[k], np_state=state, **prep_kw)
                    side.synchronize()
                    if tms[k] is not None:
                        tms[k]['stages thread: busy'] = time.perf_counter() - t_start
                    rows = prep['n'] if rng == 'numpy' else 0          # what this object's scoring pass will draw from the stream
                    st = prep['np_state']
                    if rows:            # evaluated on the draw-ahead thread, in front of the next object's draws
                        n_valid = int(transforms.valid_mask(np.asarray(ob['ob_pts'], dtype=np.float64)).sum())
                        state = (lambda st=st, nv=n_valid, r=rows: transforms.advance_choice_rows(st, nv, n_pts, r))
                    else:
                        state = st
                    q.put(('prep', prep))
        except BaseException as e:          # travels to the consumer, in order
            q.put(('error', e))

    fut = pool.submit(produce)
    results = []
    try:
        for k in range(len(objects)):
            t0 = time.perf_counter()
            kind, item = q.get()
            if kind == 'error':
                raise item
            if tms[k] is not None:
                tms[k]["waiting for this object's pre-scoring stages"] = time.perf_counter() - t0
            results.append(score_object(item, grasp_predicter, rng=rng, timings=tms[k]))
    except BaseException:
        stop[0] = True
        while fut.running() or not q.empty():       # let the stages thread finish (it may be blocked on a full queue) before the error travels on
            try:
                q.get(timeout=0.05)
            except queue.Empty:
                pass
        raise
    fut.result()
    if timings is not None:
        timings.extend(tms)
    return results
