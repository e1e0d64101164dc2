"""CPU tests (no GPU): pin the HOST-side oracle restatements to the reference's own python code.
tests/golden/host_golden.npz was produced by tests/golden/make_golden_host.py, which imports the real reference modules
(dataset_grasp, dataset_nunocs, augmentations, Utils, meshpy.sdf / sdf_file, aligning, dexnet grasp_sampler) with inert
stubs for the uninstallable packages they merely import, and runs their functions unmodified."""
import os

import numpy as np
import pytest

from oracle import aligning_ref, grasp_sampler_ref, sdf_ref
from oracle import transforms_ref as tref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'host_golden.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


def test_grasp_dataset_transform(g):
    """dataset_grasp.py:63-91, including the z>=0.1 mask, the numpy-global-RNG resample and the (x-mean)/(std+1e-15) step."""
    xyz, nrm = g['grasp_xyz'], g['grasp_normal']
    n_valid = int((xyz[:, 2] >= 0.1).sum())
    assert n_valid == len(xyz) - 5
    np.random.seed(3)
    for i, pose in enumerate(g['grasp_poses']):
        ids = tref.draw_ids(n_valid, 256)
        out = tref.grasp_transform(xyz.copy(), nrm.copy(), pose, ids, g['grasp_mean'], g['grasp_std'])['input']
        assert np.array_equal(out, g['grasp_input'][i])
    np.random.seed(4)
    ids = tref.draw_ids(n_valid, 2048)                       # fewer points than n_pts -> with replacement
    out = tref.grasp_transform(xyz.copy(), nrm.copy(), g['grasp_poses'][0], ids)['input']
    assert np.array_equal(out, g['grasp_input_replace'])


def test_nunocs_dataset_transform_and_normalize_cloud(g):
    """dataset_nunocs.py:38-65 + augmentations.py:66-75."""
    xyz, nrm = g['grasp_xyz'], g['grasp_normal']
    np.random.seed(5)
    ids = tref.draw_ids(int((xyz[:, 2] >= 0.1).sum()), 512)
    out = tref.nunocs_transform(xyz.copy(), nrm.copy(), ids)
    assert np.array_equal(out['input'], g['nunocs_input'])
    assert np.array_equal(out['keep_ids'], g['nunocs_keep_ids']) and np.array_equal(out['cloud_xyz_original'], g['nunocs_xyz_original'])
    assert np.array_equal(tref.normalize_cloud(xyz[:50].copy()), g['normalize_cloud'])
    assert np.array_equal(tref.to_homo(xyz[:4]), g['to_homo'])


def test_utils_rotation_helpers(g):
    """Utils.directionVecToRotation (Utils.py:262-290) and normalizeRotation (:172-178)."""
    for v, R in zip(g['dir2rot_dirs'], g['dir2rot']):
        assert np.allclose(grasp_sampler_ref.direction_vec_to_rotation(v.copy(), np.array([1., 0, 0])), R, atol=1e-15)
    M = g['normrot_in']
    ref = g['normrot']
    mine = M.copy(); mine[:3, :3] = grasp_sampler_ref.normalize_rotation(M[:3, :3])
    assert np.allclose(mine, ref, atol=1e-15)


def test_sdf_lookups_and_file_reader(g, tmp_path):
    """meshpy Sdf3D._signed_distance / _signed_distance_batch / is_any_points_inside (sdf.py:312-389), SdfFile._read_3d."""
    data = g['sdf_data'].astype(np.float64)
    coords = g['sdf_coords']
    data64, _, _ = sdf_ref.box_sdf_grid([-0.01, -0.004, -0.007], [0.012, 0.006, 0.003], 0.001, 5)
    assert np.allclose(sdf_ref.signed_distance(data64, coords.copy()), g['sdf_trilinear'], atol=1e-15)
    assert np.array_equal(sdf_ref.signed_distance(data64, coords.copy(), fast=True), g['sdf_fast'])
    assert np.array_equal(sdf_ref.signed_distance_batch(data.astype(np.float32), coords[None].astype(np.float32)), g['sdf_batch'])
    for c, r in zip(g['sdf_inside_coords'], g['sdf_inside']):
        assert sdf_ref.is_any_points_inside(data64, c) == bool(r)
    p = os.path.join(tmp_path, 'box.sdf')
    with open(p, 'wb') as f:
        f.write(g['sdffile_text'].tobytes())
    d, o, r = sdf_ref.read_sdf_file(p)
    assert np.array_equal(d, g['sdffile_data']) and np.array_equal(o, g['sdffile_origin']) and r == g['sdffile_res'][0]


def test_ransac_worker(g):
    """aligning.estimate9DTransform_worker (aligning.py:33-81) with the exact 4-point affine standing in for OpenCV."""
    src, dst, ids = g['ransac_src'], g['ransac_dst'], g['ransac_ids']
    n_ok = 0
    for k in range(len(ids)):
        o = aligning_ref.worker(src[ids[k]], dst[ids[k]], src, dst, 0.003, np.array([0.05] * 3), np.array([0.005, 0.005, 0.001]), np.array([1.2] * 3))
        if g['ransac_ratio'][k] < 0:
            assert o is None
        else:
            assert o is not None and abs(o[0] / len(src) - g['ransac_ratio'][k]) < 1e-12
            assert np.allclose(o[1], g['ransac_tf'][k], atol=1e-9)
            n_ok += 1
    assert n_ok > 5


def test_cone_sampler(g):
    """PointConeGraspSampler.sample_one_surface_point (grasp_sampler.py:225-298): same LAPACK eigenvectors here, so exact."""
    s = grasp_sampler_ref.ConeSampler(r_ball=0.003, hand_depth=0.02, init_bite=0.005, approach_step=0.004)
    pts, nrm = g['cone_pts'], g['cone_nrm']
    for sid, blk in zip(g['cone_ids'], g['cone_poses']):
        mine = s.sample_one_surface_point(pts[sid], nrm[sid], pts, nrm, g['cone_sphere'])
        assert mine.shape == blk.shape
        assert np.allclose(mine, blk, atol=1e-12)


def test_finger_contact_area():
    """pybullet_env/env_grasp.py:243-283 get_finger_contact_area (the real function, run through a functional open3d
    stand-in) vs oracle/affordance_ref.py."""
    from oracle import affordance_ref
    a = np.load(os.path.join(os.path.dirname(GOLD), 'affordance_golden.npz'))
    fingers = [a['finger0'], a['finger1']]
    grip_dirs = [[0, -1, 0], [0, 1, 0]]
    n_contact = 0
    for k, P in enumerate(a['poses']):
        cam_in_finger = np.linalg.inv(a['finger_mesh_in_grasp']) @ np.linalg.inv(P)
        for i in range(2):
            sp = affordance_ref.get_finger_contact_area(fingers[i], cam_in_finger, a['xyz'], grip_dirs[i], a['normal'], 0.005)
            if a['counts'][k, i] < 0:
                assert sp is None
            else:
                assert sp is not None and len(sp) == a['counts'][k, i]
                assert np.allclose(sp.mean(axis=0), a['centroid'][k, i], atol=1e-12)
                n_contact += 1
    assert n_contact > 10


def test_predicter_glue_against_real_predict_batch():
    """tests/golden/predicter_golden.npz holds the output of the REAL predicter.GraspPredicter.predict_batch (predicter.py:67-94)
    and of the network + decode lines of NunocsPredicter.predict (:135-150) run on the CPU; the oracle chain
    (transform restatement -> pointnet restatement -> softmax / argmax glue) must reproduce them."""
    import torch
    from catgrasp_amd import synth
    from oracle import pointnet_ref as oref
    p = np.load(os.path.join(os.path.dirname(GOLD), 'predicter_golden.npz'))
    sd = synth.make_state_dict('cls', 6, 10, seed=77)
    np.random.seed(123)
    xs = []
    for pose in p['poses']:
        ids = tref.draw_ids(len(p['xyz']), 2048)
        xs.append(tref.grasp_transform(p['xyz'].copy(), p['normal'].copy(), pose, ids, p['mean'], p['std'])['input'])
    logits, _ = oref.pointnet_cls_forward(sd, torch.from_numpy(np.stack(xs)).float())
    post = tref.predict_batch_post(logits.numpy())
    assert np.abs(np.array([r[2] for r in post]) - p['grasp_probs']).max() < 2e-6
    assert np.array_equal(np.array([r[0] for r in post]), p['grasp_labels'])
    assert np.abs(np.array([r[1] for r in post]) - p['grasp_conf']).max() < 2e-6
    sds = synth.make_state_dict('seg', 6, 300, seed=78)
    np.random.seed(321)
    ids = tref.draw_ids(len(p['xyz']), 8192)
    tr = tref.nunocs_transform(p['xyz'].copy(), p['normal'].copy(), ids)
    assert np.array_equal(tr['keep_ids'], p['nocs_keep_ids'])
    lg, _ = oref.pointnet_seg_forward(sds, torch.from_numpy(tr['input'][None]).float())
    coords, conf = tref.nunocs_decode(lg[0].numpy(), 100)
    clear = p['nocs_top2_gap'] > 1e-4
    assert clear.mean() > 0.99 and np.array_equal(coords[clear], p['nocs_cloud'][clear])
    assert np.abs(conf[clear[:, 2]] - p['nocs_conf_z'][clear[:, 2]]).max() < 1e-5


def test_augment_oracle_matches_the_reference_cpp():
    """oracle/augment_ref.py vs the reference's OWN C++ directionVecToRotation / augmentGraspPoses (my_cpp/common.cpp:75-153,
    compiled by oracle/build_ref.py:build_augment; outputs in tests/golden/augment_golden.npz)."""
    import os
    from oracle import augment_ref
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'augment_golden.npz'))
    R = np.array([augment_ref.direction_vec_to_rotation(d, [1, 0, 0]) for d in g['dvr_dirs']])
    assert np.abs(R - g['dvr_R']).max() <= 1e-6
    for k in range(4):
        rot, depth, step, bite = (float(v) for v in g[f'aug{k}_params'])
        mine = augment_ref.augment_grasp_poses(g[f'aug{k}_R0'], g[f'aug{k}_p'], g[f'aug{k}_sphere'], rot, depth, step, bite)
        gold = g[f'aug{k}_poses']
        assert mine.shape == gold.shape and np.abs(mine - gold).max() <= 1e-6
        # the reference itself emits 3x as many sphere rotations (its loop bound is sphere_pts.size(), reading past the matrix)
        S = len(g[f'aug{k}_sphere'])
        assert int(g[f'aug{k}_n_reference']) >= len(gold) and (S == 0) == (int(g[f'aug{k}_n_reference']) == len(gold))
    if os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libaugment_ref.so')):
        import ctypes
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libaugment_ref.so'))
        out = np.zeros(9, dtype=np.float32)
        d = np.array([0.3, -0.5, 0.8], dtype=np.float32); r = np.array([1, 0, 0], dtype=np.float32)
        lib.ref_direction_vec_to_rotation(d.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        assert np.abs(out.reshape(3, 3) - augment_ref.direction_vec_to_rotation(d, r)).max() <= 1e-6


def test_pointgroup_host_ops_oracle_matches_the_reference_cpp():
    """oracle/pointgroup_ops_ref.py: voxelization_idx / bfs_cluster vs the reference's OWN host C++ (voxelize.cpp:34-152,
    bfs_cluster.cpp:33-91 compiled by oracle/build_ref.py:build_pointgroup_host; outputs in tests/golden/pointgroup_golden.npz):
    identical maps for every mode, identical clusters incl. the queue's visit order."""
    import os
    from oracle import pointgroup_ops_ref as ref
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pointgroup_golden.npz'))
    for mode, coords in ((4, g['vox_coords']), (3, g['vox_coords']), (1, g['vox_coords']), (2, g['vox_coords']), (0, g['vox_unique_coords'])):
        oc, im, om = ref.voxelization_idx(coords, mode)
        assert np.array_equal(oc, g[f'vox_mode{mode}_output_coords']) and np.array_equal(im, g[f'vox_mode{mode}_input_map'])
        assert np.array_equal(om, g[f'vox_mode{mode}_output_map'])
    for thr in (1, 50):
        ci, co = ref.bfs_cluster(g['bfs_label'], g['bfs_idx'], g['bfs_start_len'], thr)
        assert np.array_equal(ci, g[f'bfs_thr{thr}_cluster_idxs']) and np.array_equal(co, g[f'bfs_thr{thr}_cluster_offsets'])
    assert len(g['bfs_thr1_cluster_offsets']) - 1 > 20 and len(g['bfs_thr50_cluster_offsets']) - 1 >= 4


def test_point_recover_oracle_inverts_the_golden_rule_books():
    """oracle point_recover (voxelize.cpp:182-192 -> voxelize.cu:34-48; the reference's kernel itself is the GPU-side oracle,
    tests/test_pointgroup_ops_gpu.py): on the REFERENCE's own rule books (pointgroup_golden.npz) every point receives exactly the row
    of the voxel its input map names."""
    from oracle import pointgroup_ops_ref as ref
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pointgroup_golden.npz'))
    om, im = g['vox_mode4_output_map'], g['vox_mode4_input_map']
    n = len(im)
    feats = np.random.default_rng(0).normal(size=(len(om), 6)).astype(np.float32)
    back = ref.point_recover(feats, om, n)
    assert back.shape == (n, 6) and np.array_equal(back, feats[im])
    pooled = ref.voxelize_fp(back, om, True)                 # pooling the recovered rows gives the voxel rows back (mean of equal rows)
    assert np.allclose(pooled, feats, rtol=0, atol=1e-6)


def test_c1_config_oracle_chain_against_real_predict_batch():
    """BASELINE.json configs[0] (C1) at its stated size -- one 'nut' instance, 2048-pt cloud, 256 grasp candidates, the reference's
    own CPU path: tests/golden/predicter_golden_c1.npz holds what the REAL GraspPredicter.predict_batch returned
    (make_golden_predicter.py); the oracle chain must reproduce it, and its resampling draw (n_valid == n_pts: a full permutation
    per pose) must leave numpy's global generator where the reference's loop left it."""
    import torch
    from catgrasp_amd import synth
    from oracle import pointnet_ref as oref
    p = np.load(os.path.join(os.path.dirname(GOLD), 'predicter_golden_c1.npz'))
    assert p['xyz'].shape == (2048, 3) and p['poses'].shape == (256, 4, 4)
    sd = synth.make_state_dict('cls', 6, 10, seed=79)
    np.random.seed(456)
    xs = []
    for pose in p['poses']:
        ids = tref.draw_ids(2048, 2048)
        assert len(np.unique(ids)) == 2048
        xs.append(tref.grasp_transform(p['xyz'].copy(), p['normal'].copy(), pose, ids)['input'])
    assert np.array_equal(np.random.randint(0, 2 ** 31, 4), p['rng_after'])
    with torch.no_grad():
        logits = torch.cat([oref.pointnet_cls_forward(sd, torch.from_numpy(np.stack(xs[s:s + 64])).float())[0] for s in range(0, 256, 64)])
    post = tref.predict_batch_post(logits.numpy())
    probs = np.array([r[2] for r in post])
    assert np.abs(probs - p['grasp_probs']).max() < 5e-6
    srt = np.sort(p['grasp_probs'], axis=1)
    sure = srt[:, -1] - srt[:, -2] > 1e-5
    assert sure.mean() > 0.95 and np.array_equal(np.array([r[0] for r in post])[sure], p['grasp_labels'][sure])
    assert np.abs(np.array([r[1] for r in post]) - p['grasp_conf']).max() < 5e-6


def test_oracle_kdtree_evaluation_matches_the_real_worker():
    """aligning.estimate9DTransform_worker(use_kdtree_for_eval=True) (aligning.py:63-76) run for real (cv2 / open3d substituted as
    tests/golden/make_golden_aligning_kd.py states) vs oracle/aligning_ref.worker: same accept / reject, ratios, transforms, inliers."""
    from oracle import aligning_ref as aref
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'aligning_kd_golden.npz'))
    n_acc = 0
    for si, (thr, res) in enumerate(g['settings']):
        for k in range(len(g['ids'])):
            o = aref.worker(g['src'][g['ids'][k]], g['dst'][g['ids'][k]], g['src'], g['dst'], thr, np.array([0.05] * 3), np.array([0.005, 0.005, 0.001]),
                            np.array([1.2] * 3), True, res)
            if g[f'ratio{si}'][k] < 0:
                assert o is None
                continue
            n_acc += 1
            m = np.zeros(len(g['src']), dtype=np.uint8); m[o[2]] = 1
            assert abs(o[0] - g[f'ratio{si}'][k]) < 1e-12 and np.abs(o[1] - g[f'tf{si}'][k]).max() < 1e-12 and np.array_equal(m, g[f'inliers{si}'][k])
    assert n_acc >= 40


def test_robot_gripper_loader_matches_the_real_class(tmp_path):
    """catgrasp_amd.gripper.RobotGripper.load vs the REAL dexnet RobotGripper.load on the committed fixture directory
    (tests/golden/gripper_fixture, tests/golden/gripper_golden.npz; only trimesh's OBJ parser and autolab_core's .tf parser were
    stand-ins when the golden was made): both frame orders of T_grasp_gripper.tf, the finger extents in the grasp frame incl. the
    reference's ymin / ymax convention, get_points_between_finger, every params.json key, and the SDF grids through SdfFile."""
    import shutil
    from catgrasp_amd import gripper as G
    from catgrasp_amd import sdf as sdf_mod
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, 'golden', 'gripper_golden.npz'))
    fix = os.path.join(here, 'golden', 'gripper_fixture')
    for tag, tf_name in (('fwd', 'T_grasp_gripper.tf'), ('inv', 'T_grasp_gripper_inverted.tf')):
        d = tmp_path / tag
        shutil.copytree(fix, d)
        shutil.copy(os.path.join(fix, tf_name), d / 'T_grasp_gripper.tf')
        g = G.RobotGripper.load(str(d), load_sdf=False)
        assert np.abs(g.T_grasp_gripper - gold[tag + '_T_grasp_gripper']).max() < 1e-12
        assert np.abs(g.T_grasp_gripper - gold['T_true']).max() < 1e-6          # either stored frame order yields gripper -> grasp
        assert np.abs(g.get_grasp_pose_in_gripper_base() - gold[tag + '_grasp_pose_in_gripper_base']).max() < 1e-12
        ext = np.array([g.finger_xmin, g.finger_xmax, g.finger_ymin, g.finger_ymax, g.finger_zmin, g.finger_zmax])
        assert np.abs(ext - gold[tag + '_finger_extents']).max() < 1e-12
        assert np.abs(g.finger_mesh1_in_grasp.vertices - gold[tag + '_finger_in_grasp']).max() < 1e-12
        V, F, Ve, Fe = g.filter_args()
        assert np.array_equal(V, gold[tag + '_V']) and np.array_equal(F, gold[tag + '_F'])
        assert np.array_equal(Ve, gold[tag + '_Ve']) and np.array_equal(Fe, gold[tag + '_Fe'])
        between = g.get_points_between_finger(gold['pts'])
        assert len(gold[tag + '_between']) > 10 and np.array_equal(between, gold[tag + '_between'])
        assert np.array_equal(np.array([g.hand_depth, g.init_bite, g.finger_width, g.hand_height, g.max_width, g.min_width]), gold[tag + '_params'])
        # the SDF text files as the real SdfFile.read hands them to Sdf3D (x fastest in the file, data[i][j][k])
        data, origin, res = sdf_mod.SdfFile(str(d / 'gripper_air_tight.sdf')).read_arrays()
        assert np.array_equal(data, gold[tag + '_sdf_data']) and np.array_equal(origin, gold[tag + '_sdf_origin']) and res == gold[tag + '_sdf_res'][0]
        data_e, origin_e, _ = sdf_mod.SdfFile(str(d / 'gripper_enclosed_air_tight.sdf')).read_arrays()
        assert np.array_equal(data_e, gold[tag + '_sdfe_data']) and np.array_equal(origin_e, gold[tag + '_sdfe_origin'])
