"""GPU: the per-object device pipeline (catgrasp_amd/pipeline.py) runs end to end and its pieces are mutually consistent."""
import numpy as np
import pytest

from catgrasp_amd import synth

pytestmark = pytest.mark.gpu


def test_evaluate_object_end_to_end(cuda_device):
    from catgrasp_amd import my_cpp, pipeline
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    objs = synth.make_scene(3, 1500, seed=2)
    g = synth.make_gripper()
    g['finger_vertices'] = [g['vertices'][8:16], g['vertices'][16:24]]
    g['grip_dirs'] = [[0, -1, 0], [0, 1, 0]]
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=cuda_device)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=cuda_device)
    scene_pts = np.concatenate([o['xyz'] for o in objs])
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    ob = objs[0]
    # a canonical model in the NUNOCS-scaled frame with a known pose, so the canonical-grasp branch and the affordance run
    rng = np.random.default_rng(0)
    canon_pts, canon_nrm = synth.nut_surface(3000, rng)
    canonical = {'cloud': canon_pts, 'normals': canon_nrm, 'affordance': rng.uniform(0, 1, 3000),
                 'grasps': np.linalg.inv(ob['pose']) @ synth.make_candidates(ob, 30, rng)}
    np.random.seed(1)
    timings = {}

    class FixedPose(NunocsPredicter):              # random weights cannot recover a pose: pin the RANSAC output to the truth
        def predict(self, data, ids=None):
            nocs, _, _ = self.predict_nocs(data, ids)
            self.nocs_pose = ob['pose'].copy(); self.best_ratio = 1.0
            return nocs, ob['pose'].copy()
    npred.__class__ = FixedPose
    out = pipeline.evaluate_object(ob['xyz'], ob['normal'], scene_pts, K, g, gp, npred, canonical=canonical, symmetry_tfs=[np.eye(4)],
                                   n_surface_samples=12, timings=timings)
    n = len(out['poses'])
    assert out['n_evaluated'] > 1000 and 0 < n < out['n_evaluated']
    assert out['poses'].shape == (n, 4, 4) and len(out['p_G']) == n == len(out['p_T_G'])
    assert np.all(np.diff(out['p_T_G']) <= 1e-12)                               # ranked best first
    assert np.all((out['p_G'] >= 0) & (out['p_G'] <= 0.9 + 1e-6))               # sum_k p_k k / 10, k <= 9
    assert np.allclose(out['p_T_G'], out['p_T_given_G'] * out['p_G'])
    assert set(timings) >= {'occupancy', 'nunocs+ransac', 'candidate generation', 'filterGraspPose', 'affordance', 'grasp-Q scoring'}
    # survivors really are collision free / camera facing according to an independent re-check of the filter
    bg = pipeline._background_points(scene_pts, ob['xyz'], g['diameter'], cuda_device)
    occ = my_cpp.makeOccupancyGridFromCloudScan(bg, K, 0.001)
    I4 = np.eye(4)
    codes, _, _ = my_cpp.filterGraspPoseDetailed(out['poses'][:50], [I4], I4, I4, I4, I4, g['gripper_in_grasp'], True, False, False, [0] * 7, [0] * 7,
                                                 g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], occ, 0.0005)
    assert (codes == 0).all()
    # with the IK stage (device iiwa14 solver): survivors are a subset, and each of them passes the host statement of the solver
    from oracle import iiwa_ik_ref as iiwa_ik
    cam_in_world = np.eye(4); cam_in_world[:3, :3] = [[0, -1, 0], [-1, 0, 0], [0, 0, -1]]; cam_in_world[:3, 3] = [0.55, 0.0, 0.95]
    ee_in_grasp = np.eye(4); ee_in_grasp[0, 3] = -0.15
    upper = [2.96, 2.09, 2.96, 2.09, 2.96, 2.09, 3.05]; lower = [-u for u in upper]
    np.random.seed(1)
    out_ik = pipeline.evaluate_object(ob['xyz'], ob['normal'], scene_pts, K, g, gp, npred, canonical=canonical, symmetry_tfs=[np.eye(4)],
                                      n_surface_samples=12, cam_in_world=cam_in_world, ik={'ee_in_grasp': ee_in_grasp, 'upper': upper, 'lower': lower})
    assert 0 < len(out_ik['poses']) and out_ik['n_evaluated'] == out['n_evaluated']
    ee = cam_in_world[None] @ out_ik['poses'].astype(np.float64) @ ee_in_grasp[None]
    assert iiwa_ik.ik_within_limits(ee.astype(np.float32).astype(np.float64), upper, lower).mean() > 0.999


def test_evaluate_objects_draw_ahead_equals_the_serial_loop(cuda_device):
    """the next object's NUNOCS-stage draws are made ahead of numpy's global stream while the device scores the current
    object (pipeline.evaluate_objects, run_grasp_simulation.py:188-329 order preserved) -- poses, scores, the NUNOCS transforms and the
    generator state afterwards are those of the serial per-object loop; a foreign draw in between makes the predicter drop the
    pre-drawn values instead of using stale ones."""
    from catgrasp_amd import pipeline, transforms
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    objs = synth.make_scene(4, 2300, seed=5)
    g = synth.make_gripper()
    g['finger_vertices'] = [g['vertices'][8:16], g['vertices'][16:24]]
    g['grip_dirs'] = [[0, -1, 0], [0, 1, 0]]
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=cuda_device)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=cuda_device)
    assert npred._predraw
    scene_pts = np.concatenate([o['xyz'] for o in objs])
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    rng = np.random.default_rng(0)
    canon_pts, canon_nrm = synth.nut_surface(2000, rng)
    job = [{'ob_pts': o['xyz'], 'ob_normals': o['normal'], 'symmetry_tfs': [np.eye(4)], 'nocs_pose_override': o['pose'],
            'canonical': {'cloud': canon_pts, 'normals': canon_nrm, 'affordance': rng.uniform(0, 1, 2000),
                          'grasps': np.linalg.inv(o['pose']) @ synth.make_candidates(o, 40, np.random.default_rng(k))}} for k, o in enumerate(objs)]
    kw = dict(n_surface_samples=10, rng='numpy')

    def run(**extra):
        np.random.seed(3)
        used = []
        orig = npred.predict

        def spy(data, ids=None, predrawn=None, **kw_):
            r = orig(data, ids, predrawn, **kw_)
            used.append(predrawn is not None and 'ransac id draw (exposed)' in npred.timings and npred.timings['ransac id draw (exposed)'] < 5e-3)
            return r
        npred.predict = spy
        try:
            outs = pipeline.evaluate_objects(job, scene_pts, K, g, gp, npred, **kw, **extra)
        finally:
            del npred.predict
        return outs, np.random.get_state(), used
    serial, st_serial, _ = run(draw_ahead=False)
    ahead, st_ahead, used = run(overlap='draws')
    assert transforms.same_state(st_serial, st_ahead)
    assert used[0] is False and all(used[1:])                   # objects 1.. took their pre-drawn hypothesis samples without waiting
    # the default -- the WHOLE pre-scoring half of object k+1 (occupancy, NUNOCS + RANSAC, cone sampler, filter, affordance)
    # on a second thread + stream under object k's scoring pass, every draw replayed from an explicit generator state
    tms = []
    staged, st_staged, used_s = run(timings=tms)
    assert transforms.same_state(st_serial, st_staged) and all(used_s) and len(tms) == len(job)
    assert all({'occupancy', 'nunocs+ransac', 'filterGraspPose', 'affordance', 'grasp-Q scoring'} <= set(t) for t in tms)
    for other in (ahead, staged):
        for a, b in zip(serial, other):
            assert a['n_evaluated'] == b['n_evaluated'] and len(a['poses']) > 0
            assert np.array_equal(a['poses'], b['poses']) and np.array_equal(a['p_G'], b['p_G']) and np.array_equal(a['p_T_G'], b['p_T_G'])
            assert (a['nocs_pose'] is None) == (b['nocs_pose'] is None) and (a['nocs_pose'] is None or np.array_equal(a['nocs_pose'], b['nocs_pose']))
    # the device-draw scoring mode (no numpy rows in the scoring pass: the stream passes through it untouched)
    kw_dev = dict(kw, rng='device')
    np.random.seed(3); d_serial = pipeline.evaluate_objects(job, scene_pts, K, g, gp, npred, draw_ahead=False, **kw_dev); st_d = np.random.get_state()
    np.random.seed(3); d_staged = pipeline.evaluate_objects(job, scene_pts, K, g, gp, npred, **kw_dev)
    assert transforms.same_state(st_d, np.random.get_state())
    for a, b in zip(d_serial, d_staged):      # the device draw takes a fresh seed per call: same survivors, not the same scores / order
        key = lambda r: np.ascontiguousarray(r['poses']).reshape(len(r['poses']), 16)
        assert a['n_evaluated'] == b['n_evaluated'] and np.array_equal(np.unique(key(a), axis=0), np.unique(key(b), axis=0))
    # the per-object calls of the reference loop give the same again
    np.random.seed(3)
    for ob, want in zip(job, serial):
        one = pipeline.evaluate_object(ob['ob_pts'], ob['ob_normals'], scene_pts, K, g, gp, npred, canonical=ob['canonical'], symmetry_tfs=ob['symmetry_tfs'],
                                       nocs_pose_override=ob['nocs_pose_override'], **kw)
        assert np.array_equal(one['poses'], want['poses']) and np.array_equal(one['p_T_G'], want['p_T_G'])
    assert transforms.same_state(np.random.get_state(), st_serial)
    # pre-drawn values made for another stream position are not used
    np.random.seed(3)
    fut = npred.draw_ahead(int(transforms.valid_mask(objs[0]['xyz']).sum()), np.random.get_state(), __import__('catgrasp_amd.predicter', fromlist=['x']).draw_ahead_worker())
    np.random.rand()                                            # somebody else draws in between
    st = np.random.get_state()
    a_cloud, a_pose = npred.predict({'cloud_xyz': objs[0]['xyz'], 'cloud_normal': objs[0]['normal']}, predrawn=fut.result())
    st_after = np.random.get_state()
    np.random.set_state(st)
    b_cloud, b_pose = npred.predict({'cloud_xyz': objs[0]['xyz'], 'cloud_normal': objs[0]['normal']})
    assert transforms.same_state(st_after, np.random.get_state())
    assert (a_pose is None) == (b_pose is None) and (a_pose is None or np.array_equal(a_pose, b_pose))


def test_overlapped_pick_cycle_propagates_an_error_and_stays_usable(cuda_device):
    """pipeline.evaluate_objects(overlap='stages'): an object that fails in its pre-scoring stages (a NaN point is refused at the
    cloud boundary) ends the loop with that error AFTER the objects before it were scored, the stages thread is not left blocked, and
    the next call works."""
    from catgrasp_amd import pipeline
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    objs = synth.make_scene(4, 2200, seed=6)
    g = synth.make_gripper()
    g['finger_vertices'] = [g['vertices'][8:16], g['vertices'][16:24]]
    g['grip_dirs'] = [[0, -1, 0], [0, 1, 0]]
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=cuda_device)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=cuda_device)
    scene_pts = np.concatenate([o['xyz'] for o in objs])
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    job = [{'ob_pts': o['xyz'], 'ob_normals': o['normal']} for o in objs]
    bad = [dict(j) for j in job]
    bad[2] = {'ob_pts': bad[2]['ob_pts'].copy(), 'ob_normals': bad[2]['ob_normals']}
    bad[2]['ob_pts'][5, 1] = np.nan
    np.random.seed(4)
    with pytest.raises(ValueError):
        pipeline.evaluate_objects(bad, scene_pts, K, g, gp, npred, n_surface_samples=8, rng='numpy')
    np.random.seed(4); again = pipeline.evaluate_objects(job, scene_pts, K, g, gp, npred, n_surface_samples=8, rng='numpy')
    np.random.seed(4); serial = pipeline.evaluate_objects(job, scene_pts, K, g, gp, npred, n_surface_samples=8, rng='numpy', overlap=None)
    assert len(again) == 4 and all(np.array_equal(a['poses'], b['poses']) and np.array_equal(a['p_G'], b['p_G']) for a, b in zip(again, serial))
