"""CPU tests of the blob-skipping farthest-point sampling (csrc/fps.hip, fps_blob_kernel) through its lane-by-lane host emulation
(scripts/fps_blob_lane_sim.py: the sorted-position layout, the box test against the cloud's largest running distance evaluated with the
update's own float32 operations, lane / group / slot tie detection, the slow path through `perm`, the wave records and their tie break):
whatever the kernel geometry and the cloud, the samples must equal the reference loop (pointnet2.py:54-75, restated in
oracle/pointnet_ref.py) -- the GPU tests pin the kernel itself, this pins the algorithm where no GPU is needed."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import pointnet_ref as oref

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
import fps_blob_lane_sim as sim          # noqa: E402
from fps_blob_sim import clouds          # noqa: E402


def _oracle(xyz, npoint, start):
    return oref.farthest_point_sample(torch.from_numpy(xyz[None]), npoint, torch.tensor([start]))[0].numpy()


@pytest.mark.parametrize('N,PPT,GS', [(2500, 8, 4), (5000, 16, 4), (9000, 24, 8), (20000, 40, 8)])
def test_emulated_blob_kernel_reproduces_the_reference_samples(N, PPT, GS):
    rng = np.random.default_rng(N)
    for name, xyz in clouds(N, rng):
        start = int(rng.integers(0, N))
        S = 24 if N > 10000 else 40
        got = sim.kernel(xyz, S, start, PPT=PPT, GS=GS, rng=rng, radius=True)
        assert np.array_equal(got, _oracle(xyz, S, start)), name


def test_emulated_blob_kernel_on_degenerate_clouds():
    """every running distance 0 after the first round (padding slots tie with real points and must lose), one outlier, clusters"""
    rng = np.random.default_rng(0)
    same = np.full((3000, 3), 0.25, np.float32)
    outlier = same.copy(); outlier[2999] = (1.0, 2.0, 3.0)
    centres = rng.uniform(-1, 1, (6, 3))
    clusters = (centres[rng.integers(0, 6, 3000)] + rng.normal(0, 1e-3, (3000, 3))).astype(np.float32)
    for xyz in (same, outlier, clusters):
        got = sim.kernel(xyz, 16, 5, PPT=8, GS=4, rng=rng, radius=True)
        assert np.array_equal(got, _oracle(xyz, 16, 5))


def test_box_bound_is_exact_in_float32():
    """The claim the skip rests on: for any box, any point inside it and any centre, the distance computed with the update's float32
    operations is >= the bound computed with the same operations on the box's nearest corner (each operation is monotone in |argument|)
    -- including boxes and centres at float32's edges, where a bound derived in real arithmetic would not survive the rounding."""
    rng = np.random.default_rng(1)
    f32 = np.float32
    for scale in (1e-30, 1e-6, 1.0, 1e4, 1e18):
        lo = (rng.normal(size=(20000, 3)) * scale).astype(f32)
        hi = (lo + np.abs(rng.normal(size=(20000, 3)) * scale * rng.choice([1e-7, 1e-3, 1.0], (20000, 1))).astype(f32)).astype(f32)
        p = (lo + (hi - lo) * rng.random((20000, 3)).astype(f32)).astype(f32)
        p = np.minimum(np.maximum(p, lo), hi)                      # inside the box after rounding
        c = (lo + rng.normal(size=(20000, 3)).astype(f32) * (hi - lo + f32(scale) * rng.choice([0, 1e-6, 1.0], (20000, 1)).astype(f32))).astype(f32)
        with np.errstate(over='ignore', invalid='ignore'):
            d = p - c
            d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            q = np.maximum(np.maximum(lo - c, c - hi), f32(0))
            lb = (q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]
        ok = np.isfinite(d) & np.isfinite(lb)
        assert d.dtype == f32 and lb.dtype == f32 and ok.sum() > 10000
        assert np.all(d[ok] >= lb[ok])
