"""CPU tests of the occupancy ray cast oracle (cr_make_occupancy_grid, the restatement of my_cpp/common.cpp:324-431 + octomap's
castRay that csrc/occupancy.hip is held bit-equal to on the GPU) against an independent geometric decision procedure
(oracle/occupancy_exact.py: slab test of every occupied leaf, no voxel walk) -- in BOTH directions, outside a stated epsilon band."""
import numpy as np

from catgrasp_amd import synth
from oracle import collision_oracle as co
from oracle import occupancy_exact as oe


def _emitted_mask(queries, emitted):
    table = {q.tobytes() for q in np.ascontiguousarray(emitted, dtype=np.float32)}
    return np.array([q.tobytes() in table for q in np.ascontiguousarray(queries, dtype=np.float32)])


def _check(pts, res, min_decidable=0.9):
    pts = np.asarray(pts, dtype=np.float32)
    got = co.make_occupancy_grid(pts, res)
    q, max_range = oe.lattice(pts, res)
    assert len(got) <= len(q)
    emitted = _emitted_mask(q, got)
    assert emitted.sum() == len(got), 'the oracle emitted a point that is not on the query lattice'
    occ, ok = oe.decide(pts, res, q, max_range)
    assert ok.mean() >= min_decidable, f'only {ok.mean():.3f} of the queries are decidable'
    wrong = ok & (occ != emitted)
    assert not wrong.any(), (f'{wrong.sum()} of {ok.sum()} decidable queries differ: over-reported {int((wrong & emitted).sum())}, '
                             f'missed {int((wrong & ~emitted).sum())}')
    return emitted, occ, ok


def test_ray_cast_equals_slab_geometry_on_a_clutter_scan():
    objs = synth.make_scene(2, 700, seed=5)
    pts = np.concatenate([o['xyz'] for o in objs])
    for res in (0.002, 0.003):
        emitted, occ, ok = _check(pts, res)
        assert 50 < emitted.sum() < len(emitted)           # both outcomes are exercised
        assert (occ & ok).sum() > 50 and (~occ & ok).sum() > 1000


def test_ray_cast_on_synthetic_walls_and_single_points():
    # a wall facing the sensor: everything behind it (within the lattice) is occupied, everything in front is free
    g = np.linspace(-0.02, 0.02, 41)
    wall = np.stack(np.meshgrid(g, g, [0.5], indexing='ij'), -1).reshape(-1, 3)
    emitted, occ, ok = _check(wall, 0.002, min_decidable=0.7)
    q, _ = oe.lattice(wall.astype(np.float32), 0.002)
    assert not emitted[q[:, 2] < 0.497].any() and emitted[(q[:, 2] > 0.5045) & (np.abs(q[:, :2]) < 0.015).all(1)].all()
    # one point: only lattice points inside its shadow cone are occupied
    emitted, occ, ok = _check(np.array([[0.01, -0.02, 0.6]]), 0.002)
    assert 0 < emitted.sum() < 40
    # an over-reporting mutant (every lattice point behind ANY leaf's distance) must be caught by the two-sided comparison
    pts = np.concatenate([o['xyz'] for o in synth.make_scene(1, 300, seed=2)]).astype(np.float32)
    q, max_range = oe.lattice(pts, 0.002)
    occ, ok = oe.decide(pts, 0.002, q, max_range)
    mutant = np.sqrt((q.astype(np.float64) ** 2).sum(1)) >= np.sqrt(((oe.occupied_leaves(pts, 0.002) + 0.5) ** 2).sum(1)).min() * 0.002
    assert (ok & (mutant != occ)).sum() > 100


def test_cast_ray_variants_of_the_sensitivity_study_are_live():
    """oracle/occupancy_sensitivity.py reports 0 flipped lattice points on the C3 background clouds for every alternative reading of
    octomap::castRay; this shows that each switch does change the walk where its case arises (so 0 is a measurement, not a dead flag)."""
    from oracle import collision_oracle as co
    res = 0.001
    try:
        # a tie of tMax on the diagonal: `<` steps y first, `<=` steps x first -> different leaves are visited on the way
        d = np.float32([1, 1, 0]) / np.float32(np.sqrt(2.0))
        only_x = np.float32([[0.0015, 0.0005, 0.0005]])             # leaf (1, 0, 0)
        co.set_occupancy_variant()
        hit0, _ = co.cast_ray(only_x, res, d, 1.0)
        co.set_occupancy_variant(tie=1)
        hit1, end1 = co.cast_ray(only_x, res, d, 1.0)
        assert (hit0, hit1) == (False, True) and np.allclose(end1, [0.0015, 0.0005, 0.0005])
        # an occupied leaf whose centre lies just beyond maxRange
        far = np.float32([[0.0105, 0.0005, 0.0005]])                # leaf (10, 0, 0), centre at 10.5 mm
        co.set_occupancy_variant()
        assert co.cast_ray(far, res, np.float32([1, 0, 0]), 0.0104)[0] is False
        co.set_occupancy_variant(range_last=1)
        assert co.cast_ray(far, res, np.float32([1, 0, 0]), 0.0104)[0] is True
        # leaf centres in float arithmetic: NOT a degree of freedom -- (k + 0.5) and res are floats, their double product is exact, so
        # rounding it once IS the float product
        diff = 0
        for k in range(1, 400):
            p = np.float32([[(k + 0.5) * res, 0.0005, 0.0005]])
            co.set_occupancy_variant()
            _, a = co.cast_ray(p, res, np.float32([1, 0, 0]), 10.0)
            co.set_occupancy_variant(fcoord=1)
            _, b = co.cast_ray(p, res, np.float32([1, 0, 0]), 10.0)
            diff += int(a[0] != b[0])
        assert diff == 0
    finally:
        co.set_occupancy_variant()
