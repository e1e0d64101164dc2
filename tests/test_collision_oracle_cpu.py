"""CPU tests (no GPU) of the collision oracle (oracle/collision_ref.c).  PARITY UNPINNED: FCL/octomap are
absent, so instead of golden vectors the restatement is checked against independent formulations:
octomap's published key formula in numpy float64; an INDEPENDENT triangle/box decision procedure (polygon clipping in exact
rationals / float64, oracle/tribox_exact.py -- no separating axes) that the float32 SAT must match in BOTH directions; properties
FCL's mesh-vs-octree semantics imply (monotone in the voxel set, invariant under re-triangulation, surface-only); and
control-flow invariants of filterGraspPose (my_cpp/common.cpp:156-321)."""
import numpy as np

from catgrasp_amd import synth
from oracle import collision_oracle as co

I4 = np.eye(4)


def test_voxel_keys_follow_octomap_formula():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (4000, 3)).astype(np.float32)
    pts = np.concatenate([pts, np.array([[99999, 0, 0], [0, 16.3841, 0], [16.3839, -16.384, 0.0]], dtype=np.float32)])
    for res in (0.0005, 0.001):
        resf = np.float64(np.float32(res))
        k = np.floor(pts.astype(np.float64) * (1.0 / resf)).astype(np.int64) + 32768     # OcTreeBaseImpl::coordToKeyChecked
        ok = ((k >= 0) & (k < 65536)).all(axis=1)
        uniq = np.unique(k[ok], axis=0) - 32768
        got = co.voxelize(pts, res)
        assert np.array_equal(got, uniq)
    assert co.voxelize(np.ones((1, 3)) * 99999, 0.0005).shape == (0, 3)


def test_tri_box_sat_equals_exact_clipping_on_dyadic_inputs_both_directions():
    """Coordinates on a 1/16 lattice: every product and sum of the float32 SAT is exact, so it must agree EXACTLY with the
    rational clipping oracle -- including all touching configurations (the lattice makes them common: ~15 % of the cases have a
    vertex on a face plane, edges through cube edges, triangles coplanar with a face).  A predicate that over-reports (or
    under-reports) contact fails here."""
    from oracle import tribox_exact as tx
    rng = np.random.default_rng(11)
    n_hit = n_miss = n_touch = n_mutant_caught = 0
    for _ in range(8000):
        h = rng.integers(1, 9) / 16.0
        c = rng.integers(-16, 17, 3) / 16.0
        scale = rng.choice([4, 16, 40])
        a, b, d = (c + rng.integers(-scale, scale + 1, 3) / 16.0 for _ in range(3))
        if _ % 4 == 0:          # contact family: 1..3 vertices exactly on a face plane of the cube
            ax, side = rng.integers(0, 3), rng.choice([-1.0, 1.0])
            for v in (a, b, d)[:rng.integers(1, 4)]:
                v[ax] = c[ax] + side * h
        want = tx.tri_box_intersect(c, h, a, b, d, exact=True)
        got = co.tri_box_overlap(c, h, a, b, d)
        assert got == want, (c, h, a, b, d, got, want)
        n_hit += want; n_miss += not want
        n_mutant_caught += co.tri_box_overlap(c, h + 1 / 64, a, b, d) != want      # an over-reporting predicate (cube inflated by 1/64)
        if want and not tx.tri_box_intersect(c, h * (1 - 2.0 ** -10), a, b, d, exact=True):
            n_touch += 1                                    # intersects the closed cube but not a slightly smaller one: contact only
    assert n_hit > 1000 and n_miss > 1000 and n_touch > 100, (n_hit, n_miss, n_touch)
    assert n_mutant_caught > 20          # ... would have failed the equality above on this many cases: the test is two-sided
    # hand-made contact cases: vertex on a face, edge along a cube edge, coplanar with a face, point contact at a corner
    z = np.zeros(3)
    for a, b, d, want in (((1, 0, 0), (2, 1, 0), (2, -1, 0), True), ((1, 1, -3), (1, 1, 3), (5, 5, 0), True),
                          ((-3, -3, 1), (3, -3, 1), (0, 3, 1), True), ((1, 1, 1), (2, 1, 1), (1, 2, 2), True),
                          ((1.0625, 0, 0), (2, 1, 0), (2, -1, 0), False), ((-3, -3, 1.0625), (3, -3, 1.0625), (0, 3, 1.0625), False)):
        a, b, d = np.array(a, float), np.array(b, float), np.array(d, float)
        assert tx.tri_box_intersect(z, 1.0, a, b, d) == want == co.tri_box_overlap(z, 1.0, a, b, d)


def test_tri_box_sat_agrees_with_clipping_on_random_float32_inputs_outside_an_epsilon_band():
    """10^5 random + near-grazing float32 cases against float64 clipping, two-sided: if the triangle intersects the cube shrunk by
    eps the SAT must report overlap; if it misses the cube grown by eps the SAT must report disjoint.  eps = 4e-6 of the scene
    scale (float32 rounding of the 13 axis tests; coordinates are O(1))."""
    from oracle import tribox_exact as tx
    rng = np.random.default_rng(1)
    eps = 4e-6
    n_in = n_out = n_band = 0
    for i in range(100000):
        h = np.float32(rng.uniform(0.05, 0.6))
        c = rng.normal(0, 0.3, 3).astype(np.float32)
        if i % 3 == 0:       # grazing family: a triangle in a plane at distance ~h from the centre along a random axis
            ax = rng.integers(0, 3)
            a, b, d = (rng.normal(0, 1.0, 3) for _ in range(3))
            off = c[ax] + (h + rng.normal(0, 3e-6)) * rng.choice([-1, 1])
            a[ax] = off + rng.normal(0, 1e-6); b[ax] = off + rng.normal(0, 1e-6); d[ax] = off + rng.normal(0, 1e-6)
        else:
            a, b, d = (rng.normal(0, 1.0, 3) for _ in range(3))
        a, b, d = a.astype(np.float32), b.astype(np.float32), d.astype(np.float32)
        got = co.tri_box_overlap(c, h, a, b, d)
        if tx.tri_box_intersect(c, float(h) - eps, a, b, d, exact=False):
            assert got, ('under-reports', c, h, a, b, d)
            n_in += 1
        elif not tx.tri_box_intersect(c, float(h) + eps, a, b, d, exact=False):
            assert not got, ('over-reports', c, h, a, b, d)
            n_out += 1
        else:
            n_band += 1
    assert n_in > 20000 and n_out > 20000 and 0 < n_band < 20000, (n_in, n_out, n_band)


def test_tri_box_symmetry_and_containment():
    rng = np.random.default_rng(2)
    for _ in range(500):
        c = rng.normal(0, 0.2, 3); a, b, d = (rng.normal(0, 0.6, 3) for _ in range(3))
        r = co.tri_box_overlap(c, 0.3, a, b, d)
        # vertex order does not matter
        assert r == co.tri_box_overlap(c, 0.3, b, d, a) == co.tri_box_overlap(c, 0.3, d, b, a)
        # growing the box never turns an overlap into a miss
        if r:
            assert co.tri_box_overlap(c, 0.6, a, b, d)
    # triangle entirely inside / huge triangle through the box / far away
    assert co.tri_box_overlap(np.zeros(3), 1.0, np.array([0.1, 0, 0]), np.array([0, 0.1, 0]), np.array([0, 0, 0.1]))
    assert co.tri_box_overlap(np.zeros(3), 0.1, np.array([-9, -9, 0.0]), np.array([9, -9, 0.0]), np.array([0, 9, 0.0]))
    assert not co.tri_box_overlap(np.zeros(3), 0.1, np.array([-9, -9, 0.2]), np.array([9, -9, 0.2]), np.array([0, 9, 0.2]))


def _run(P, sym, nocs, g, pts, bg, dirf, adj):
    return co.filter_grasp_pose(P, sym, nocs, I4, I4, I4, g['gripper_in_grasp'], dirf, 0, adj, g['vertices'], g['faces'],
                                g['enclosed_vertices'], g['enclosed_faces'], pts, bg, 0.0005)


def test_filter_control_flow_invariants():
    objs = synth.make_scene(4, 1500, 1)
    g = synth.make_gripper()
    bg = synth.background_points(objs, 0, g['diameter'])
    P = synth.make_candidates(objs[0], 300, np.random.default_rng(3))
    c0, p0, n0 = _run(P, [I4], I4, g, objs[0]['xyz'], bg, 1, 0)
    c1, p1, n1 = _run(P, [I4], I4, g, objs[0]['xyz'], bg, 1, 1)
    assert set(np.unique(c0)) <= {0, 1, 3, 4} and set(np.unique(c1)) <= {0, 1, 3}
    assert np.array_equal(c0 == 1, c1 == 1)                               # approach-dir test is independent of adjust
    assert ((c0 == 0) <= (c1 == 0)).all()                                  # nudge 0 is tried first
    assert np.array_equal(n1[c0 == 0], np.zeros((c0 == 0).sum(), dtype=np.int8))
    assert (n1[c1 != 0] == -1).all() and (n0[c0 == 0] == 0).all()
    # surviving poses: rotation columns normalised, last row 0 0 0 1, z of approach >= 0
    keep = p1[c1 == 0]
    assert np.allclose(np.linalg.norm(keep[:, :3, :3], axis=1), 1, atol=1e-6)
    assert (keep[:, 2, 0] >= 0).all() and np.array_equal(keep[:, 3], np.tile([0, 0, 0, 1], (len(keep), 1)))
    assert (p1[c1 != 0] == 0).all()
    # nudged poses moved along their own y axis by exactly the float32 step
    moved = np.nonzero((c1 == 0) & (n1 > 0))[0]
    assert len(moved) > 0
    for e in moved:
        step = np.float32([0, 0.001, -0.001, 0.002, -0.002][n1[e]])
        base = _run(P[e:e + 1], [I4], I4, g, np.ones((1, 3)) * 99999, np.ones((1, 3)) * 99999, 0, 0)[1][0]
        assert np.allclose(p1[e][:3, 3] - base[:3, 3], step * base[:3, 1], atol=1e-7)
    # no approach-dir filter => code 1 never appears; empty clouds => nothing collides
    c2, _, _ = _run(P, [I4], I4, g, np.ones((1, 3)) * 99999, np.ones((1, 3)) * 99999, 0, 0)
    assert (c2 == 0).all()
    # symmetry expansion multiplies the evaluation count, in i-major order
    S = np.eye(4); S[:3, :3] = [[-1, 0, 0], [0, -1, 0], [0, 0, 1]]
    c3, p3, _ = _run(P[:10], [I4, S], I4, g, objs[0]['xyz'], bg, 1, 0)
    assert len(c3) == 20 and np.array_equal(c3[0::2], c0[:10])


def test_mesh_voxels_collide_basic():
    V, F = synth.box_mesh([-0.01, -0.01, -0.01], [0.01, 0.01, 0.01])
    inside_surface = np.array([[0.01, 0.0, 0.0]], dtype=np.float32)        # on the +x face
    deep_inside = np.array([[0.0, 0.0, 0.0]], dtype=np.float32)            # strictly inside: BVH-vs-octree is surface-only
    far = np.array([[0.1, 0.1, 0.1]], dtype=np.float32)
    pose = np.eye(4, dtype=np.float32)
    assert co.mesh_voxels_collide(V, F, pose, co.voxelize(inside_surface, 0.001), 0.001)
    assert not co.mesh_voxels_collide(V, F, pose, co.voxelize(deep_inside, 0.001), 0.001)
    assert not co.mesh_voxels_collide(V, F, pose, co.voxelize(far, 0.001), 0.001)
    pose[:3, 3] = [0.1, 0.1, 0.09]
    assert co.mesh_voxels_collide(V, F, pose, co.voxelize(far, 0.001), 0.001)


def test_sdf_oracle_trilinear_reproduces_linear_fields_and_lattice_values():
    """oracle/sdf_ref.py: trilinear interpolation is exact on a (tri)linear field and at lattice points."""
    from oracle import sdf_ref
    rng = np.random.default_rng(0)
    i, j, k = np.meshgrid(np.arange(9), np.arange(8), np.arange(7), indexing='ij')
    data = 0.3 * i - 0.2 * j + 0.05 * k + 1.0
    c = rng.uniform(0, 6, (3, 200))
    ref = 0.3 * c[0] - 0.2 * c[1] + 0.05 * c[2] + 1.0
    assert np.abs(sdf_ref.signed_distance(data, c) - ref).max() < 1e-12
    lat = np.stack([rng.integers(0, 9, 50), rng.integers(0, 8, 50), rng.integers(0, 7, 50)]).astype(float)
    assert np.allclose(sdf_ref.signed_distance(data, lat), data[lat[0].astype(int), lat[1].astype(int), lat[2].astype(int)])
    assert np.array_equal(sdf_ref.signed_distance(data, np.array([[0.5], [1.5], [2.5]]), fast=True), data[0, 2, 2:3])  # half-even
    g, origin, res = sdf_ref.box_sdf_grid([0, 0, 0], [0.01, 0.02, 0.005], 0.001, 5)
    assert g.shape == (30, 30, 30) and g.min() < 0 < g.max()


def _subdivide(V, F):
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    n0 = len(V)
    mids = np.stack([(a + b) / 2, (b + c) / 2, (c + a) / 2], axis=1).reshape(-1, 3)
    i = n0 + 3 * np.arange(len(F))
    F2 = np.concatenate([np.stack([F[:, 0], i, i + 2], 1), np.stack([i, F[:, 1], i + 1], 1), np.stack([i + 2, i + 1, F[:, 2]], 1),
                         np.stack([i, i + 1, i + 2], 1)]).astype(np.int32)
    return np.concatenate([V, mids]).astype(np.float32), F2


def test_mesh_vs_voxels_properties_implied_by_fcl_semantics():
    """What ANY correct mesh-vs-octree collision must satisfy, whatever its narrow phase: (i) monotone in the voxel set,
    (ii) invariant under re-triangulating the same surface (flip a quad's diagonal / split every triangle in four), (iii)
    surface-only: leaves strictly inside a closed mesh, away from its surface, do not collide, (iv) rigid motion of both."""
    rng = np.random.default_rng(5)
    res = 0.0005
    V, F = synth.box_mesh([-0.008, -0.004, -0.002], [0.008, 0.004, 0.002])       # dyadic-free metric box, closed surface
    V = V.astype(np.float32)
    F_flip = F.copy()
    for q in range(0, 12, 2):        # the two triangles of each face share a diagonal: use the other diagonal
        t0, t1 = F[q], F[q + 1]
        quad = [t0[0], t0[1], t0[2]] + [v for v in t1 if v not in t0]
        shared = [v for v in t0 if v in t1]
        others = [v for v in quad if v not in shared]
        F_flip[q] = [others[0], shared[0], others[1]]; F_flip[q + 1] = [others[0], others[1], shared[1]]
    V4, F4 = _subdivide(V, F)
    n_hit = 0
    for trial in range(300):
        T = np.eye(4, dtype=np.float32); T[:3, :3] = synth.random_rotation(rng); T[:3, 3] = rng.uniform(-0.004, 0.004, 3)
        pts = (rng.uniform(-0.012, 0.012, (rng.integers(1, 40), 3))).astype(np.float32)
        keys = co.voxelize(pts, res)
        r = co.mesh_voxels_collide(V, F, T, keys, res)
        n_hit += r
        assert r == co.mesh_voxels_collide(V, F_flip, T, keys, res)                       # (ii) other diagonal
        assert r == co.mesh_voxels_collide(V4, F4, T, keys, res)                          # (ii) 1:4 subdivision
        sub = keys[rng.random(len(keys)) < 0.5]
        if co.mesh_voxels_collide(V, F, T, sub, res):                                     # (i) subset hit => superset hit
            assert r
        extra = np.concatenate([keys, co.voxelize(rng.uniform(-0.012, 0.012, (5, 3)).astype(np.float32), res)])
        if r:
            assert co.mesh_voxels_collide(V, F, T, np.unique(extra, axis=0), res)
        # (iii) points strictly inside the posed box, > one voxel diagonal from every face, never collide on their own
        inner = rng.uniform([-0.0065, -0.0025, -0.0008], [0.0065, 0.0025, 0.0008], (20, 3))
        inner = (inner @ T[:3, :3].T.astype(np.float64) + T[:3, 3]).astype(np.float32)
        assert not co.mesh_voxels_collide(V, F, T, co.voxelize(inner, res), res)
    assert 60 < n_hit < 280


def test_sensitivity_of_the_predicate_to_its_unpinnable_details():
    """FCL / octomap cannot be run here, so the exposure is MEASURED instead (oracle/collision_sensitivity.py; the full C3 batch is in
    profiles/r4_collision_sensitivity.json).  The parity oracle decides by float64 polygon clipping.  On a 6,000-evaluation cut of the
    C3 batch: the float32 separating-axis test the HIP kernel runs, FCL's own leaf boxes (16 float halvings of the root BV), and
    libccd's MPR with FCL's default tolerance (the narrow phase FCL's default solver runs for a triangle / leaf-box pair, restated)
    must reproduce the oracle's codes / nudges / poses up to a handful of evaluations, and a +-1 um change of the cube's half edge may
    move at most 0.1 % of the codes.  The variant switch must leave the default path untouched."""
    from oracle import collision_sensitivity as cs
    batch, objs, gripper, nocs, cats, n_total = cs.c3_batch(per_replica=6000)
    cs.set_variant()
    base = cs.run_batch(batch, objs, gripper, nocs, cats)
    assert n_total == 6000 and len(base[0]) == n_total and set(np.unique(base[0])) <= {0, 1, 3, 4}
    try:
        for name in ('sat_f32', 'fcl_halving', 'fcl_halving+sat_f32'):
            cs.set_variant(*cs.VARIANTS[name])
            c, n, p = cs.run_batch(batch, objs, gripper, nocs, cats)
            assert np.array_equal(c, base[0]) and np.array_equal(n, base[1]) and np.array_equal(p, base[2]), name
        for name in ('mpr_libccd', 'fcl_halving+mpr_libccd', 'grow_1um', 'shrink_1um'):
            cs.set_variant(*cs.VARIANTS[name])
            c, n, _ = cs.run_batch(batch, objs, gripper, nocs, cats)
            assert (c != base[0]).sum() <= 6 and ((n != base[1]) & (c == base[0])).sum() <= 6, name
    finally:
        cs.set_variant()
    again = cs.run_batch(batch, objs, gripper, nocs, cats)
    assert all(np.array_equal(a, b) for a, b in zip(again, base))
    # grazing pairs (closest approach within +-2 um of contact): the float32 SAT and the float64 clipping disagree on at most 1 in 1,000
    # of THESE, libccd's MPR (tolerance 1e-6) on at most 1 %, the FCL-halved leaf boxes on ~1 % (an ulp of 0.6 m = 6e-8 m)
    keys, a, b, e = cs.grazing_pairs(4000, seed=1)
    g0 = cs.grazing_decisions(keys, a, b, e, 0, 0.0, 0)
    assert 0.3 < g0.mean() < 0.7
    assert (cs.grazing_decisions(keys, a, b, e, 0, 0.0, 1) != g0).mean() <= 0.001
    assert (cs.grazing_decisions(keys, a, b, e, 0, 0.0, 2) != g0).mean() < 0.01
    assert (cs.grazing_decisions(keys, a, b, e, 1, 0.0, 0) != g0).mean() < 0.03


def test_mpr_restatement_agrees_with_clipping_away_from_contact():
    """cr_tri_box_mpr (libccd's portal refinement, restated from the published algorithm) is a correct intersection test: on random
    triangle / cube pairs that are not within a micrometre of contact it decides exactly like the float64 clipping and like the exact
    rational clipping of tribox_exact.py."""
    import ctypes
    from oracle import tribox_exact
    rng = np.random.default_rng(7)
    l = co.lib()
    n = 6000
    c = rng.uniform(-0.01, 0.01, (n, 3)).astype(np.float32); h3 = np.full(3, 0.00025, dtype=np.float32)
    A = (c + rng.normal(0, 0.0015, (n, 3))).astype(np.float32); B = (A + rng.normal(0, 0.003, (n, 3))).astype(np.float32)
    D = (A + rng.normal(0, 0.003, (n, 3))).astype(np.float32)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    clip = np.array([l.cr_tri_box_clip64(P(c[i]), P(h3), P(A[i]), P(B[i]), P(D[i])) for i in range(n)], dtype=bool)
    mpr = np.array([l.cr_tri_box_mpr(P(c[i]), P(h3), P(A[i]), P(B[i]), P(D[i])) for i in range(n)], dtype=bool)
    assert 100 < clip.sum() < n - 100 and np.array_equal(clip, mpr)
    for i in np.flatnonzero(clip)[:40].tolist() + np.flatnonzero(~clip)[:40].tolist():
        exact = tribox_exact.tri_box_intersect(c[i], h3[0], A[i], B[i], D[i], exact=True)
        assert exact == bool(clip[i])


def test_general_half_extent_sat_equals_the_cube_sat():
    """cr_tri_box_overlap_h3 (per-axis half extents, used by the sensitivity variants) == cr_tri_box_overlap when the box is a cube."""
    import ctypes
    rng = np.random.default_rng(4)
    l = co.lib()
    n = 20000
    c = rng.normal(0, 0.01, (n, 3)).astype(np.float32); h = np.float32(0.00025)
    tri = (c[:, None, :] + rng.normal(0, 0.0004, (n, 3, 3))).astype(np.float32)
    h3 = np.array([h, h, h], dtype=np.float32)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    agree = 0
    for i in range(n):
        a = l.cr_tri_box_overlap(P(c[i]), ctypes.c_float(h), P(tri[i, 0]), P(tri[i, 1]), P(tri[i, 2]))
        b = l.cr_tri_box_overlap_h3(P(c[i]), P(h3), P(tri[i, 0]), P(tri[i, 1]), P(tri[i, 2]))
        agree += int(a == b)
    assert agree == n


def _sat17(P, Q):
    """closed triangle / closed triangle by 17 separating axes in float64 (numpy): an independent check of the oracle's
    segment-through-triangle formulation (and the arithmetic plan of the HIP kernel, csrc/collision_pairs.hip)."""
    P = np.asarray(P, np.float64).reshape(3, 3); Q = np.asarray(Q, np.float64).reshape(3, 3)
    Q = Q - P[0]; P = P - P[0]
    ep = [P[(i + 1) % 3] - P[i] for i in range(3)]; eq = [Q[(i + 1) % 3] - Q[i] for i in range(3)]
    npn, nqn = np.cross(ep[0], ep[1]), np.cross(eq[0], eq[1])
    axes = [npn, nqn] + [np.cross(a, b) for a in ep for b in eq] + [np.cross(npn, a) for a in ep] + [np.cross(nqn, b) for b in eq]
    for L in axes:
        a, b = P @ L, Q @ L
        if a.max() < b.min() or b.max() < a.min():
            return False
    return True


def test_tri_tri_oracle_known_answers_and_agreement_with_separating_axes():
    t = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    assert co.tri_tri_overlap(t, t)                                                             # itself
    assert co.tri_tri_overlap(t, [[0.2, 0.2, -1], [0.2, 0.2, 1], [2, 2, 0.5]])                  # an edge pierces the interior
    assert not co.tri_tri_overlap(t, [[0.2, 0.2, 0.1], [0.3, 0.2, 1], [2, 2, 0.5]])             # above the plane
    assert not co.tri_tri_overlap(t, [[2, 2, -1], [2, 2, 1], [3, 3, 0]])                        # crosses the plane outside the triangle
    assert co.tri_tri_overlap(t, t + np.float32([0.25, 0.25, 0]))                               # coplanar, overlapping
    assert not co.tri_tri_overlap(t, t + np.float32([3, 0, 0]))                                 # coplanar, apart
    assert co.tri_tri_overlap(t, t * 0.2 + np.float32([0.1, 0.1, 0]))                           # coplanar, contained
    assert co.tri_tri_overlap(t, [[1, 0, 0], [2, 0, 1], [2, 1, -1]])                            # touching in one vertex (closed sets)
    assert not co.tri_tri_overlap(t, t * 0.2 + np.float32([0.1, 0.1, 0.01]))                    # parallel planes
    rng = np.random.default_rng(5)
    n_hit = 0
    for _ in range(6000):
        P = rng.normal(0, 1, (3, 3)).astype(np.float32); Q = (rng.normal(0, 1, (3, 3)) + rng.normal(0, 0.8, 3)).astype(np.float32)
        got = co.tri_tri_overlap(P, Q)
        assert got == _sat17(P, Q)
        n_hit += int(got)
    assert 500 < n_hit < 5000


def test_box_box_oracle_known_answers_and_agreement_with_separating_axes():
    I = np.eye(3, dtype=np.float32)
    assert co.box_box_overlap([0, 0, 0], 1.0, [1.5, 0, 0], 1.0, I) and not co.box_box_overlap([0, 0, 0], 1.0, [2.5, 0, 0], 1.0, I)
    assert co.box_box_overlap([0, 0, 0], 1.0, [0.1, 0, 0], 0.2, I) and co.box_box_overlap([0, 0, 0], 0.2, [0.1, 0, 0], 1.0, I)     # containment, both ways
    c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    assert co.box_box_overlap([0, 0, 0], 1.0, [2.3, 0, 0], 1.0, Rz) and not co.box_box_overlap([0, 0, 0], 1.0, [2.5, 0, 0], 1.0, Rz)  # a corner reaches 1.414
    assert not co.box_box_overlap([0, 0, 0], 1.0, [2.2, 2.2, 0], 1.0, Rz)                       # only an edge-edge axis separates... (face normals do too here)

    def sat15(ca, ha, cb, hb, R):
        ca, cb, R = np.asarray(ca, np.float64), np.asarray(cb, np.float64), np.asarray(R, np.float64).reshape(3, 3)
        t = cb - ca
        axes = [np.eye(3)[i] for i in range(3)] + [R[:, j] for j in range(3)] + [np.cross(np.eye(3)[i], R[:, j]) for i in range(3) for j in range(3)]
        for L in axes:
            ra = ha * np.abs(L).sum()
            rb = hb * sum(abs(L @ R[:, j]) for j in range(3))
            if abs(t @ L) > ra + rb:
                return False
        return True
    rng = np.random.default_rng(8)
    n_hit = 0
    for _ in range(4000):
        R = synth.random_rotation(rng).astype(np.float32)
        cb = rng.normal(0, 1.3, 3).astype(np.float32)
        hb = float(np.float32(rng.uniform(0.3, 1.2)))
        got = co.box_box_overlap([0, 0, 0], 1.0, cb, hb, R)
        assert got == sat15([0, 0, 0], 1.0, cb, hb, R)
        n_hit += int(got)
    assert 800 < n_hit < 3500


def test_mesh_mesh_and_voxels_voxels_oracle_basic():
    g = synth.make_gripper()
    assert co.mesh_mesh_collide(g['vertices'], g['faces'], np.eye(4), g['enclosed_vertices'], g['enclosed_faces'], np.eye(4))
    T = np.eye(4); T[0, 3] = 1.0
    assert not co.mesh_mesh_collide(g['vertices'], g['faces'], np.eye(4), g['enclosed_vertices'], g['enclosed_faces'], T)
    pts = np.random.default_rng(0).uniform(0, 0.01, (300, 3)).astype(np.float32)
    k = co.voxelize(pts, 0.0005)
    assert co.voxels_voxels_collide(k, 0.0005, k, 0.0005, np.eye(4))
    assert not co.voxels_voxels_collide(k, 0.0005, k, 0.0005, T)
    # the same key set at another resolution is another place: (key + 0.5) res
    assert co.voxels_voxels_collide(k, 0.0005, co.voxelize(pts, 0.001), 0.001, np.eye(4))

