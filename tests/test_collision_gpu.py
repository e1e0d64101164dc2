"""GPU parity (bit-exact): HIP filterGraspPose / CollisionManager / voxelisation vs the C oracle
(oracle/collision_ref.c), same seeded inputs.  Integer/boolean outputs must be identical; surviving
poses must be bit-identical float32."""
import numpy as np
import pytest
import torch

from catgrasp_amd import synth
from oracle import collision_oracle as co

pytestmark = pytest.mark.gpu
I4 = np.eye(4)


def _scene(seed=0, n_obj=6, pts=2500):
    objs = synth.make_scene(n_obj, pts, seed)
    g = synth.make_gripper()
    bg = synth.background_points(objs, 0, g['diameter'])
    return objs, g, bg


def _args(P, sym, nocs, g, obj_pts, bg, dirf, adj, res=0.0005):
    return (P, sym, nocs, I4, I4, I4, g['gripper_in_grasp'], dirf, False, adj, [0] * 7, [0] * 7, g['vertices'], g['faces'],
            g['enclosed_vertices'], g['enclosed_faces'], obj_pts, bg, res)


def _oracle(P, sym, nocs, g, obj_pts, bg, dirf, adj, res=0.0005):
    return co.filter_grasp_pose(P, sym, nocs, I4, I4, I4, g['gripper_in_grasp'], dirf, 0, adj, g['vertices'], g['faces'],
                                g['enclosed_vertices'], g['enclosed_faces'], obj_pts, bg, res)


def _check(dev, ora):
    codes, poses, nudge = dev
    ocodes, oposes, onudge = ora
    assert np.array_equal(codes, ocodes), f'{(codes != ocodes).sum()} of {len(codes)} codes differ'
    assert np.array_equal(nudge, onudge)
    assert np.array_equal(poses.view(np.uint32), oposes.view(np.uint32)), 'surviving poses are not bit-identical'


def test_voxelize_matches_octomap_keys(cuda_device):
    from catgrasp_amd import my_cpp
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.2, 0.8, (5000, 3)).astype(np.float32)
    # exact voxel boundaries, negatives, duplicates, out-of-tree points (ignored by updateNode)
    extra = np.array([[0.0005 * 7, -0.0005 * 3, 0.0], [0.0005 * 7, -0.0005 * 3, 0.0], [-1e-9, 1e-9, 0.6], [99999, 99999, 99999],
                      [16.3839, -16.3840, 0], [16.3841, 0, 0], [0, -16.39, 0]], dtype=np.float32)
    pts = np.concatenate([pts, extra])
    for res in (0.0005, 0.001, 0.0025):
        keys = my_cpp.voxelize(pts, res).cpu().numpy()
        okeys = co.voxelize(pts, res)
        assert keys.shape[0] == okeys.shape[0]
        assert np.array_equal(keys[:, :3].astype(np.int32), okeys)
    assert my_cpp.voxelize(np.ones((1, 3)) * 99999, 0.0005).shape[0] == 0     # generate_grasp.py:97 sentinel cloud


@pytest.mark.parametrize('adjust', [False, True])
def test_filter_cone_sampler_path(cuda_device, adjust):
    """grasp_sampler.py:216 call shape: symmetry=[I], nocs_pose=I, approach-dir filter on."""
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(0)
    P = synth.make_candidates(objs[0], 600, np.random.default_rng(1))
    dev = my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, g, objs[0]['xyz'], bg, True, adjust))
    ora = _oracle(P, [I4], I4, g, objs[0]['xyz'], bg, 1, int(adjust))
    _check(dev, ora)
    assert (dev[0] == 0).sum() > 0 and (dev[0] == 3).sum() > 0      # both outcomes exercised
    lst = my_cpp.filterGraspPose(*_args(P, [I4], I4, g, objs[0]['xyz'], bg, True, adjust), False)
    assert len(lst) == int((ora[0] == 0).sum())
    for m, e in zip(lst, np.nonzero(ora[0] == 0)[0]):
        assert m.dtype == np.float32 and np.array_equal(m, ora[1][e])


def test_filter_nocs_transfer_path_symmetry_and_scale(cuda_device):
    """grasp_sampler.py:345 call shape: 12 nut symmetries, a 9-D (anisotropically scaled) nocs_pose,
    adjust_collision_pose=True, no approach-dir filter."""
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(2)
    rng = np.random.default_rng(5)
    obj = objs[0]
    T = obj['pose']
    scale = np.diag([0.016, 0.016, 0.006, 1.0])
    nocs_pose = T @ scale                                   # canonical (unit NUNOCS cube) -> camera
    # canonical grasps: object-frame candidates mapped into the canonical frame
    P_cam = synth.make_candidates(obj, 40, rng)
    P_can = np.linalg.inv(nocs_pose) @ P_cam
    sym = []
    for xa in (0, np.pi):
        for za in np.arange(0, 360, 60) / 180 * np.pi:
            Rx = np.array([[1, 0, 0], [0, np.cos(xa), -np.sin(xa)], [0, np.sin(xa), np.cos(xa)]])
            Rz = np.array([[np.cos(za), -np.sin(za), 0], [np.sin(za), np.cos(za), 0], [0, 0, 1]])
            S = np.eye(4); S[:3, :3] = Rx @ Rz
            sym.append(S)
    dev = my_cpp.filterGraspPoseDetailed(*_args(P_can, sym, nocs_pose, g, obj['xyz'], bg, False, True))
    ora = _oracle(P_can, sym, nocs_pose, g, obj['xyz'], bg, 0, 1)
    assert len(dev[0]) == 40 * 12
    _check(dev, ora)


@pytest.mark.parametrize('cls,kind,n_sym', [('hnm', 'screw', 2), ('screw', 'screw', 72), ('nut', 'nut', 12)])
def test_filter_category_symmetry_sets(cuda_device, cls, kind, n_sym):
    """The per-category symmetry expansion of the NOCS-transfer sampler (Utils.py:79-94 -> grasp_sampler.py:345)."""
    from catgrasp_amd import my_cpp, transforms
    objs = synth.make_scene(4, 2000, seed=11, kind=kind)
    g = synth.make_gripper()
    bg = synth.background_points(objs, 1, g['diameter'])
    obj = objs[1]
    sym = transforms.get_symmetry_tfs(cls)
    assert len(sym) == n_sym and all(np.allclose(S[:3, :3] @ S[:3, :3].T, np.eye(3)) for S in sym)
    nocs_pose = obj['pose'] @ np.diag([0.01, 0.01, 0.05, 1.0])
    P_can = np.linalg.inv(nocs_pose) @ synth.make_candidates(obj, 24, np.random.default_rng(3))
    dev = my_cpp.filterGraspPoseDetailed(*_args(P_can, sym, nocs_pose, g, obj['xyz'], bg, True, True))
    ora = _oracle(P_can, sym, nocs_pose, g, obj['xyz'], bg, 1, 1)
    assert len(dev[0]) == 24 * n_sym
    _check(dev, ora)


def test_filter_edge_cases(cuda_device):
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(4, n_obj=3, pts=800)
    P = synth.make_candidates(objs[0], 64, np.random.default_rng(7))
    sentinel = np.ones((1, 3)) * 99999
    # offline generate_grasp.py:97 path: sentinel clouds => nothing collides
    dev = my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, g, sentinel, sentinel, True, False))
    ora = _oracle(P, [I4], I4, g, sentinel, sentinel, 1, 0)
    _check(dev, ora)
    assert set(np.unique(dev[0])) <= {0, 1}
    # empty pose list
    codes, poses, nudge = my_cpp.filterGraspPoseDetailed(*_args(np.zeros((0, 4, 4)), [I4], I4, g, objs[0]['xyz'], bg, True, True))
    assert codes.shape == (0,) and poses.shape == (0, 4, 4)
    assert my_cpp.filterGraspPose(*_args([], [I4], I4, g, objs[0]['xyz'], bg, True, True), False) == []
    # wrong shapes raise instead of exit(1)
    with pytest.raises(ValueError):
        my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, dict(g, vertices=g['vertices'][:, :2]), objs[0]['xyz'], bg, True, False))
    # coarser resolution + a finely tessellated gripper (> one LDS triangle chunk)
    V, F = g['vertices'], g['faces']
    for _ in range(2):     # 36 -> 144 -> 576 triangles
        nv = len(V); newV = [V]; newF = []
        for f in F:
            a, b, c = V[f[0]], V[f[1]], V[f[2]]
            m = np.stack([(a + b) / 2, (b + c) / 2, (c + a) / 2]).astype(np.float32)
            i0 = nv; nv += 3; newV.append(m)
            newF += [[f[0], i0, i0 + 2], [i0, f[1], i0 + 1], [i0 + 2, i0 + 1, f[2]], [i0, i0 + 1, i0 + 2]]
        V = np.concatenate(newV).astype(np.float32); F = np.array(newF, dtype=np.int32)
    g2 = dict(g, vertices=V, faces=F)
    dev = my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, g2, objs[0]['xyz'], bg, True, False, 0.001))
    ora = _oracle(P, [I4], I4, g2, objs[0]['xyz'], bg, 1, 0, 0.001)
    _check(dev, ora)


def test_collision_manager_api(cuda_device):
    """my_cpp.CollisionManager surface (collision_manager.h:60-65)."""
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(6, n_obj=2, pts=1500)
    cm = my_cpp.CollisionManager()
    gid = cm.registerMesh(g['vertices'], g['faces'])
    cid = cm.registerPointCloud(objs[0]['xyz'].astype(np.float32), 0.0005)
    assert (gid, cid) == (0, 1)
    okeys = co.voxelize(objs[0]['xyz'], 0.0005)
    P = synth.make_candidates(objs[0], 24, np.random.default_rng(9))
    n_hit = 0
    for p in P:
        pose = (p @ g['gripper_in_grasp']).astype(np.float32)
        cm.setTransform(pose, gid)
        got = cm.isAnyCollision()
        exp = co.mesh_voxels_collide(g['vertices'], g['faces'], pose, okeys, 0.0005)
        assert got == exp
        n_hit += int(got)
    assert 0 < n_hit < len(P)
    with pytest.raises(ValueError):
        cm.setTransform(np.eye(3), gid)
    with pytest.raises(ValueError):
        cm.registerMesh(np.zeros((4, 2)), np.zeros((1, 3), dtype=np.int32))
    # a rigidly posed cloud (collision_manager.cpp:81-91 lets any object be posed): the same leaf boxes seen from another frame, i.e.
    # the oracle's answer for the mesh pose inv(cloud pose) . mesh pose
    Tc = np.eye(4); Tc[:3, :3] = synth.random_rotation(np.random.default_rng(3)); Tc[:3, 3] = [0.02, -0.01, 0.03]
    cm.setTransform(Tc.astype(np.float32), cid)
    n_hit = 0
    for p in P:
        pose_w = (Tc @ p @ g['gripper_in_grasp']).astype(np.float32)                  # the gripper moved along with the cloud ...
        cm.setTransform(pose_w, gid)
        rel = (np.linalg.inv(Tc.astype(np.float32).astype(np.float64)) @ pose_w.astype(np.float64)).astype(np.float32)
        got = cm.isAnyCollision()
        assert got == co.mesh_voxels_collide(g['vertices'], g['faces'], rel, okeys, 0.0005)
        n_hit += int(got)
    assert 0 < n_hit < len(P)                                                          # ... so the verdicts are those of the unposed scene
    with pytest.raises(ValueError):                     # a cloud can be moved, not scaled
        cm.setTransform(np.diag([2.0, 2.0, 2.0, 1.0]).astype(np.float32), cid)
    with pytest.raises(ValueError):                     # ... nor sheared: a unit-determinant matrix that is not a rotation
        S = np.eye(4, dtype=np.float32); S[0, 1] = 0.5
        cm.setTransform(S, cid)


def _soup(rng, n, extent, size):
    """n random triangles of edge ~size scattered in a cube of side `extent`."""
    c = rng.uniform(-extent / 2, extent / 2, size=(n, 1, 3))
    V = (c + rng.normal(0, size, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    return V, np.arange(3 * n, dtype=np.int32).reshape(n, 3)


def _rigid(rng, t_scale):
    T = np.eye(4); T[:3, :3] = synth.random_rotation(rng); T[:3, 3] = rng.normal(0, t_scale, 3)
    return T.astype(np.float32)


def test_collision_manager_tests_every_pair_mesh_mesh_and_cloud_cloud(cuda_device):
    """collision_manager.cpp:93-111 loops over EVERY pair of registered objects: mesh / mesh (triangle against triangle) and cloud /
    cloud (leaf cube against leaf cube, each set in its own pose) against the oracle's float64 segment-through-triangle / clipping
    predicates, on configurations from clearly apart to deeply interpenetrating; then the managers' any-pair logic."""
    from catgrasp_amd import my_cpp
    rng = np.random.default_rng(17)
    # ---- mesh / mesh
    VA, FA = _soup(rng, 300, 0.05, 0.004)
    VB, FB = _soup(rng, 500, 0.05, 0.004)
    n_hit = 0
    for k in range(40):
        cm = my_cpp.CollisionManager()
        a = cm.registerMesh(VA, FA); b = cm.registerMesh(VB, FB)
        Ta = _rigid(rng, 0.01)
        Tb = _rigid(rng, 0.01); Tb[0, 3] += np.float32(0.003 * k)      # slides B away from A
        cm.setTransform(Ta, a); cm.setTransform(Tb, b)
        got = cm.isAnyCollision()
        assert got == co.mesh_mesh_collide(VA, FA, Ta, VB, FB, Tb), k
        n_hit += int(got)
    assert 0 < n_hit < 40
    # the gripper meshes of the path against each other (closed surfaces, thousands of triangles when subdivided)
    g = synth.make_gripper()
    for shift, expect in ((0.0, True), (0.5, False)):
        cm = my_cpp.CollisionManager()
        a = cm.registerMesh(g['vertices'], g['faces']); b = cm.registerMesh(g['enclosed_vertices'], g['enclosed_faces'])
        T = np.eye(4, dtype=np.float32); T[0, 3] = shift
        cm.setTransform(T, b)
        assert cm.isAnyCollision() == expect == co.mesh_mesh_collide(g['vertices'], g['faces'], np.eye(4), g['enclosed_vertices'], g['enclosed_faces'], T)
    # coplanar pairs: overlapping, and apart inside the same plane (the six in-plane axes decide)
    tri = np.array([[0, 0, 0], [0.01, 0, 0], [0, 0.01, 0]], np.float32); F1 = np.array([[0, 1, 2]], np.int32)
    for dx, expect in ((0.004, True), (0.02, False)):
        cm = my_cpp.CollisionManager()
        cm.registerMesh(tri, F1); cm.registerMesh(tri + np.float32([dx, 0.001, 0]), F1)
        assert cm.isAnyCollision() == expect == co.mesh_mesh_collide(tri, F1, np.eye(4), tri + np.float32([dx, 0.001, 0]), F1, np.eye(4))
    # a small triangle strictly inside a big one's plane region but lifted off it: parallel planes
    cm = my_cpp.CollisionManager(); cm.registerMesh(tri, F1); cm.registerMesh(tri * 0.2 + np.float32([0.002, 0.002, 0.0005]), F1)
    assert cm.isAnyCollision() is False
    # ---- cloud / cloud
    objs, _, bg = _scene(3, n_obj=2, pts=1200)
    pa, pb = objs[0]['xyz'].astype(np.float32), objs[1]['xyz'].astype(np.float32)
    ka, kb = co.voxelize(pa, 0.0005), co.voxelize(pb, 0.001)
    n_hit = 0
    ca, cb = pa.mean(0), pb.mean(0)
    for k in range(30):
        cm = my_cpp.CollisionManager()
        a = cm.registerPointCloud(pa, 0.0005); b = cm.registerPointCloud(pb, 0.001)
        Ta = np.eye(4, dtype=np.float32)
        Tb = _rigid(rng, 0.0); Tb[:3, 3] = (ca - Tb[:3, :3] @ cb + rng.normal(0, 0.0004 * k, 3)).astype(np.float32)    # B's cloud dropped onto A's, then apart
        if k % 3 == 0:
            Ta = _rigid(rng, 0.01); Tb = (Ta.astype(np.float64) @ Tb.astype(np.float64)).astype(np.float32)
        cm.setTransform(Ta, a); cm.setTransform(Tb, b)
        rel = (np.linalg.inv(Ta.astype(np.float64)) @ Tb.astype(np.float64)).astype(np.float32)
        got = cm.isAnyCollision()
        assert got == co.voxels_voxels_collide(ka, 0.0005, kb, 0.001, rel), k
        n_hit += int(got)
    assert 0 < n_hit < 30
    # the same cloud twice, unposed: every cube meets itself
    cm = my_cpp.CollisionManager(); cm.registerPointCloud(pa, 0.0005); cm.registerPointCloud(pa, 0.0005)
    assert cm.isAnyCollision() is True
    # ---- three objects: the loop reports a hit in ANY pair (collision_manager.cpp:95-108)
    far = np.eye(4, dtype=np.float32); far[:3, 3] = [5, 5, 5]
    cm = my_cpp.CollisionManager()
    m0 = cm.registerMesh(VA, FA); m1 = cm.registerMesh(VB, FB); c0 = cm.registerPointCloud(pa, 0.0005)
    cm.setTransform(far, m1); far2 = far.copy(); far2[:3, 3] = [-5, 5, 5]; cm.setTransform(far2, m0)
    assert cm.isAnyCollision() is False                       # all three apart
    cm.setTransform(np.eye(4, dtype=np.float32), m1)
    Tm = np.eye(4, dtype=np.float32); Tm[:3, 3] = ca          # mesh 1 onto the cloud
    cm.setTransform(Tm, m1)
    assert cm.isAnyCollision() == co.mesh_voxels_collide(VB, FB, Tm, ka, 0.0005)
    cm.setTransform(far, m1); cm.setTransform(far, m0)        # the two meshes onto each other, far from the cloud
    assert cm.isAnyCollision() is True
    # empty objects never collide
    cm = my_cpp.CollisionManager(); cm.registerMesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)); cm.registerMesh(VA, FA)
    assert cm.isAnyCollision() is False


def test_tri_box_predicate_random_and_grazing(cuda_device):
    """The SAT predicate itself on random and near-touching configurations, via single-voxel clouds."""
    from catgrasp_amd import my_cpp
    rng = np.random.default_rng(11)
    res = 0.001
    V = np.array([[0, 0, 0], [0.004, 0, 0], [0, 0.003, 0]], dtype=np.float32)
    F = np.array([[0, 1, 2]], dtype=np.int32)
    vox = np.array([[0.0105, 0.0205, 0.0305]], dtype=np.float32)        # centre of key (10,20,30)
    keys = co.voxelize(vox, res)
    dkeys = my_cpp.voxelize(vox, res)
    poses = []
    for _ in range(4000):
        R = synth.random_rotation(rng)
        t = vox[0] + rng.normal(0, 0.0012, 3)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t - R @ np.array([0.0013, 0.001, 0])
        poses.append(T)
    # grazing: triangle plane exactly on / one ulp off a voxel face
    for dz in (0.0, 1e-9, -1e-9, 6e-8, -6e-8):
        T = np.eye(4); T[:3, 3] = [0.0095, 0.0195, 0.031 + dz]
        poses.append(T)
    poses = np.array(poses, dtype=np.float32)
    import ctypes
    from catgrasp_amd import _lib as L
    dev = torch.device('cuda:0')
    dV = torch.from_numpy(V).to(dev); dF = torch.from_numpy(F).to(dev)
    dP = torch.from_numpy(poses.reshape(-1, 16)).to(dev)
    out = torch.zeros((len(poses),), dtype=torch.uint8, device=dev)
    L.check(L.lib().cg_mesh_voxels_collide(L._p(dV), L._p(dF), ctypes.c_int(1), L._p(dP), ctypes.c_long(len(poses)), L._p(dkeys),
                                           ctypes.c_int(dkeys.shape[0]), ctypes.c_float(res), L._p(out), L._stream()), 'collide')
    got = out.cpu().numpy().astype(bool)
    exp = np.array([co.mesh_voxels_collide(V, F, p, keys, res) for p in poses])
    assert np.array_equal(got, exp)
    assert 0.05 < exp.mean() < 0.95


def test_augment_grasp_poses(cuda_device):
    """my_cpp.augmentGraspPoses (common.cpp:118-153) vs the numpy/SVD restatement; float tolerance 1e-5."""
    from catgrasp_amd import my_cpp
    from oracle import augment_ref
    rng = np.random.default_rng(13)
    R0 = synth.random_rotation(rng)
    p = np.array([0.01, -0.02, 0.6])
    sph = rng.normal(size=(7, 3)); sph /= np.linalg.norm(sph, axis=1, keepdims=True)
    sph[0] = [1, 0, 0]                       # parallel to the reference axis -> identity branch (:79-82)
    sph[1] = [-1, 1e-7, 0]                   # anti-parallel: |v| < 1e-5 also returns identity in the reference
    got = np.array(my_cpp.augmentGraspPoses(R0, p, sph, 30.0, 0.04, 0.002, 0.005))
    ref = augment_ref.augment_grasp_poses(R0, p, sph, 30.0, 0.04, 0.002, 0.005)
    assert got.shape == ref.shape == ((1 + 7 * 6) * 20, 4, 4) and got.dtype == np.float32
    assert np.abs(got - ref).max() <= 1e-5
    assert np.abs(np.linalg.det(got[:, :3, :3]) - 1).max() < 1e-5
    with pytest.raises(ValueError):
        my_cpp.augmentGraspPoses(np.eye(4), p, sph, 30.0, 0.04, 0.002, 0.005)
    assert my_cpp.augmentGraspPoses(R0, p, np.zeros((0, 3)), 30.0, 0.04, 0.002, 0.005)[0].shape == (4, 4)
    # ... and against the REFERENCE's own C++ (common.cpp:118-153 compiled by oracle/build_ref.py; tests/golden/augment_golden.npz)
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'augment_golden.npz'))
    for k in range(4):
        rot, depth, step, bite = (float(v) for v in g[f'aug{k}_params'])
        got = np.array(my_cpp.augmentGraspPoses(g[f'aug{k}_R0'], g[f'aug{k}_p'], g[f'aug{k}_sphere'], rot, depth, step, bite))
        assert got.shape == g[f'aug{k}_poses'].shape and np.abs(got - g[f'aug{k}_poses']).max() <= 1e-5


def test_make_occupancy_grid_from_cloud_scan(cuda_device):
    """my_cpp.makeOccupancyGridFromCloudScan (common.cpp:324-431): identical lattice-point set to the C oracle."""
    from catgrasp_amd import my_cpp
    objs = synth.make_scene(3, 1500, 5)
    pts = np.concatenate([o['xyz'] for o in objs[:2]]).astype(np.float32)
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1.0]])
    for res in (0.001, 0.002):
        got = my_cpp.makeOccupancyGridFromCloudScan(pts, K, res)
        ref = co.make_occupancy_grid(pts, res)
        assert got.dtype == np.float32 and got.shape == ref.shape and len(ref) > 100
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # every scan point has an emitted lattice point within ~1.5 voxels (the surface itself is "at or behind")
    from scipy.spatial import cKDTree
    d, _ = cKDTree(my_cpp.makeOccupancyGridFromCloudScan(pts, K, 0.001)).query(pts)
    assert d.max() < 0.002
    assert my_cpp.makeOccupancyGridFromCloudScan(np.zeros((0, 3)), K, 0.001).shape == (0, 3)
    with pytest.raises(ValueError):
        my_cpp.makeOccupancyGridFromCloudScan(np.zeros((4, 2)), K, 0.001)


@pytest.mark.skipif(not co.ikfast_available(), reason='oracle/_ref/libikfast_ref.so is built from /root/reference in the build container')
def test_filter_with_ik_stage(cuda_device):
    """filter_ik=True (common.cpp:214-226): pre-IK device stage -> host IK (the reference's own IKFast solver, oracle/_ref)
    -> collision stage; codes incl. the IK rejections must equal the oracle driven by the same solver."""
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(8, n_obj=3, pts=1200)
    P = synth.make_candidates(objs[0], 200, np.random.default_rng(17))
    upper = [2.96, 2.09, 2.96, 2.09, 2.96, 2.09, 3.05]            # iiwa14 joint limits (rad)
    lower = [-u for u in upper]
    cam_in_world = np.eye(4); cam_in_world[:3, :3] = [[0, -1, 0], [-1, 0, 0], [0, 0, -1]]; cam_in_world[:3, 3] = [0.55, 0.0, 0.95]
    ee_in_grasp = np.eye(4); ee_in_grasp[0, 3] = -0.15
    my_cpp.set_ik_solver(co.ikfast_within_limits)
    try:
        dev = my_cpp.filterGraspPoseDetailed(P, [I4], I4, I4, cam_in_world, ee_in_grasp, g['gripper_in_grasp'], True, True, False, upper, lower,
                                             g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], objs[0]['xyz'], bg, 0.0005)
    finally:
        my_cpp.set_ik_solver(None)
    up = np.array(upper); lo = np.array(lower)

    def ik_cb(ee_ptr, _user):
        ee = np.ctypeslib.as_array(ee_ptr, shape=(16,)).copy()
        return int(co.ikfast_within_limits(ee.reshape(1, 4, 4), up, lo)[0])
    ora = co.filter_grasp_pose(P, [I4], I4, I4, cam_in_world, ee_in_grasp, g['gripper_in_grasp'], 1, 1, 0, g['vertices'], g['faces'],
                               g['enclosed_vertices'], g['enclosed_faces'], objs[0]['xyz'], bg, 0.0005, ik_fn=ik_cb)
    _check(dev, ora)
    assert (ora[0] == 2).sum() > 0 and (ora[0] == 0).sum() > 0
    # default solver (no callback registered): the device closed-form iiwa14 IK must reproduce the IKFast-driven result
    dev2 = my_cpp.filterGraspPoseDetailed(P, [I4], I4, I4, cam_in_world, ee_in_grasp, g['gripper_in_grasp'], True, True, False, upper, lower,
                                          g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], objs[0]['xyz'], bg, 0.0005)
    _check(dev2, ora)
    with pytest.raises(ValueError):                # limits are mandatory with filter_ik
        my_cpp.filterGraspPoseDetailed(P, [I4], I4, I4, cam_in_world, ee_in_grasp, g['gripper_in_grasp'], True, True, False, None, None,
                                       g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], objs[0]['xyz'], bg, 0.0005)


def test_broad_phase_grid_is_result_neutral(cuda_device):
    """The mesh-frame grid broad phase (cg_mesh_grid) must not change a single code/pose: exhaustive vs accelerated vs oracle,
    on the box gripper, a finely tessellated gripper (2304 triangles), symmetric + sheared poses (which fall back)."""
    from catgrasp_amd import my_cpp
    objs, g, bg = _scene(10, n_obj=4, pts=2000)
    P = synth.make_candidates(objs[0], 300, np.random.default_rng(21))
    V, F = g['vertices'], g['faces']
    for _ in range(3):     # 36 -> 2304 triangles
        nv = len(V); newV = [V]; newF = []
        for f in F:
            a, b, c = V[f[0]], V[f[1]], V[f[2]]
            m = np.stack([(a + b) / 2, (b + c) / 2, (c + a) / 2]).astype(np.float32)
            i0 = nv; nv += 3; newV.append(m)
            newF += [[f[0], i0, i0 + 2], [i0, f[1], i0 + 1], [i0 + 2, i0 + 1, f[2]], [i0, i0 + 1, i0 + 2]]
        V = np.concatenate(newV).astype(np.float32); F = np.array(newF, dtype=np.int32)
    g_fine = dict(g, vertices=V, faces=F)
    for gg, adj in ((g, True), (g_fine, False)):
        a = my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, gg, objs[0]['xyz'], bg, True, adj), accel=True)
        b = my_cpp.filterGraspPoseDetailed(*_args(P, [I4], I4, gg, objs[0]['xyz'], bg, True, adj), accel=False)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    ora = _oracle(P, [I4], I4, g_fine, objs[0]['xyz'], bg, 1, 0)
    _check(a, ora)
    # anisotropic nocs_pose: sheared gripper poses (sigma_min < 0.5 for some) -> exhaustive fallback inside the same launch
    T = objs[0]['pose']
    nocs_pose = T @ np.diag([0.03, 0.012, 0.004, 1.0])
    P_can = np.linalg.inv(nocs_pose) @ P[:60]
    S = np.eye(4); S[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    a = my_cpp.filterGraspPoseDetailed(*_args(P_can, [I4, S], nocs_pose, g, objs[0]['xyz'], bg, False, True), accel=True)
    ora = _oracle(P_can, [I4, S], nocs_pose, g, objs[0]['xyz'], bg, 0, 1)
    _check(a, ora)


def test_device_narrow_phase_equals_the_exact_clipping_oracle(cuda_device):
    """The HIP narrow phase pinned to the INDEPENDENT decision procedure (oracle/tribox_exact.py: rational polygon clipping, no
    separating axes), not only to its C twin: one-triangle meshes on a 1/32 lattice, translated by lattice vectors (exact in
    float32), against a single occupied leaf of edge 1/16 -- every contact configuration included."""
    import ctypes
    from catgrasp_amd import _lib as L
    from oracle import tribox_exact as tx
    rng = np.random.default_rng(4)
    res = 1.0 / 16
    key = np.array([[3, -2, 1, 0]], dtype=np.int16)
    centre = (key[0, :3].astype(np.float64) + 0.5) * res
    keys = torch.from_numpy(key).to(cuda_device)
    F = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=cuda_device)
    n_hit = n_total = n_touch = 0
    for _ in range(40):
        tri = rng.integers(-6, 7, (3, 3)) / 32.0
        shifts = rng.integers(-4, 5, (200, 3)) / 32.0 + np.round(centre * 32) / 32.0
        poses = np.tile(np.eye(4, dtype=np.float32), (200, 1, 1)); poses[:, :3, 3] = shifts
        V = torch.from_numpy(tri.astype(np.float32)).to(cuda_device)
        out = torch.zeros((200,), dtype=torch.uint8, device=cuda_device)
        L.check(L.lib().cg_mesh_voxels_collide(L._p(V), L._p(F), ctypes.c_int(1), L._p(torch.from_numpy(poses.reshape(200, 16)).to(cuda_device)),
                                               ctypes.c_long(200), L._p(keys), ctypes.c_int(1), ctypes.c_float(res), L._p(out), L._stream()),
                'cg_mesh_voxels_collide')
        got = out.cpu().numpy().astype(bool)
        for i in range(200):
            a, b, d = tri + shifts[i]
            want = tx.tri_box_intersect(centre, res / 2, a, b, d, exact=True)
            assert got[i] == want, (tri, shifts[i], got[i], want)
            n_hit += want; n_total += 1
            n_touch += want and not tx.tri_box_intersect(centre, res / 2 * (1 - 2.0 ** -10), a, b, d, exact=True)
    assert 500 < n_hit < n_total - 500 and n_touch > 100, (n_hit, n_total, n_touch)


def test_filter_plan_of_several_calls_equals_the_calls_and_the_oracle(cuda_device):
    """cg_filter_grasp_pose_multi (my_cpp.FilterPlan): the filter calls of several objects -- both call shapes, different symmetry sets,
    nocs poses, nudge flags and voxel sets, an empty segment, an object without background, partial symmetry ranges -- as ONE launch
    sequence: every evaluation's code / nudge / pose is bit-identical to its own filter_on_device call, and (for two of the segments)
    to the C oracle."""
    from catgrasp_amd import my_cpp, transforms, workload
    objs, g, _ = _scene(4, n_obj=5, pts=1800)
    dev = cuda_device
    rng = np.random.default_rng(8)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 16)).to(dev)
    scenes, bgs = [], []
    for k, ob in enumerate(objs):
        bg = synth.background_points(objs, k, g['diameter']) if k != 3 else np.ones((1, 3)) * 99999      # object 3: the sentinel cloud
        bgs.append(bg)
        scenes.append(my_cpp.GripperScene(g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005, dev))
    eye = f32(np.eye(4)[None])
    rows, host = [], []
    for k, ob in enumerate(objs):
        cat = ('nut', 'hnm', 'screw')[k % 3]
        sym = np.stack(transforms.get_symmetry_tfs(cat))
        nocs = workload.scene_nocs_pose(ob)
        can = np.linalg.inv(nocs) @ synth.make_candidates(ob, 40 + 7 * k, rng, g['hand_depth'], g['init_bite'])
        cone = synth.make_candidates(ob, 150 + 31 * k, rng, g['hand_depth'], g['init_bite'])
        j0, j1 = (0, len(sym)) if k != 2 else (5, 61)                # a partial symmetry range, as a shard boundary produces
        rows.append((scenes[k], f32(can), f32(sym[j0:j1]), nocs, I4, True)); host.append((k, can, sym[j0:j1], nocs, True))
        rows.append((scenes[k], f32(cone), eye, I4, I4, k % 2 == 1)); host.append((k, cone, np.eye(4)[None], I4, k % 2 == 1))
        if k == 1:
            rows.append((scenes[k], f32(np.zeros((0, 4, 4))), eye, I4, I4, False)); host.append((k, np.zeros((0, 4, 4)), np.eye(4)[None], I4, False))
    plan = my_cpp.FilterPlan(rows)
    assert plan.E == sum(len(P) * len(S) for _, P, S, _, _ in host)
    for keep in (True, False):
        codes, poses, nudge = plan.run(g['gripper_in_grasp'], True, keep_rejected_pose=keep)
        for (k, P, S, nocs, adj), first, count in zip(host, plan.firsts, plan.counts):
            c1, p1, n1 = my_cpp.filter_on_device(scenes[k], f32(P), f32(S), nocs, I4, I4, I4, g['gripper_in_grasp'], True, False, adj, keep_rejected_pose=keep)
            sl = slice(first, first + count)
            assert torch.equal(codes[sl], c1) and torch.equal(nudge[sl], n1) and torch.equal(poses[sl].view(torch.int32), p1.view(torch.int32))
    assert len(set(codes.cpu().tolist())) >= 3                        # keeps, direction rejects and collisions all occur
    # two segments against the C oracle directly (keep_rejected_pose=False: rejected poses are zeros, as the oracle returns them)
    for q in (0, 3):
        k, P, S, nocs, adj = host[q]
        ora = _oracle(P, list(S), nocs, g, objs[k]['xyz'], bgs[k], 1, int(adj))
        sl = slice(plan.firsts[q], plan.firsts[q] + plan.counts[q])
        _check((codes[sl].cpu().numpy(), poses[sl].cpu().numpy(), nudge[sl].cpu().numpy()), ora)
    # scenes without broad-phase grids: the sequence runs on the exhaustive kernel alone -- same results
    plain = [my_cpp.GripperScene(g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], objs[k]['xyz'], bgs[k], 0.0005, dev, accel=False)
             for k in (0, 1)]
    rows_plain = [(plain[k],) + rows[q][1:] for q, (k, _, _, _, _) in enumerate(host) if k in (0, 1)]
    sel = [q for q, (k, _, _, _, _) in enumerate(host) if k in (0, 1)]
    c2, p2, n2 = my_cpp.FilterPlan(rows_plain).run(g['gripper_in_grasp'], True)
    want = torch.cat([codes[plan.firsts[q]:plan.firsts[q] + plan.counts[q]] for q in sel])
    assert torch.equal(c2, want) and int(c2.numel()) == sum(plan.counts[q] for q in sel)
    # a plan must not mix grippers
    other = synth.make_gripper(subdivisions=1)
    sc2 = my_cpp.GripperScene(other['vertices'], other['faces'], other['enclosed_vertices'], other['enclosed_faces'], objs[0]['xyz'], bgs[0], 0.0005, dev)
    with pytest.raises(ValueError):
        my_cpp.FilterPlan([rows[0], (sc2,) + rows[1][1:]])
