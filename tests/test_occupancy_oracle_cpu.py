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
