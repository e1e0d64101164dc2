"""GPU parity of the PointNet++ primitives (pointnet2.py:14-149) vs the CPU oracle restatement
(oracle/pointnet_ref.py, itself checked against the imported reference in tests/golden).
Index outputs are exact; ball-query membership may differ from the torch-CPU evaluation only for points
whose expanded squared distance is within float rounding (1e-6 relative band) of r^2."""
import numpy as np
import pytest
import torch

from oracle import pointnet_ref as oref

pytestmark = pytest.mark.gpu


def _cloud(B, N, seed, scale=0.05, offset=(0.0, 0.0, 0.6)):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.normal(0, scale, (B, N, 3)) + np.array(offset)).astype(np.float32))


def test_square_distance(cuda_device):
    from catgrasp_amd import pointnet2 as p2
    src, dst = _cloud(2, 70, 1), _cloud(2, 333, 2)
    got = p2.square_distance(src.to(cuda_device), dst.to(cuda_device)).cpu()
    ref = oref.square_distance(src, dst)
    assert got.shape == (2, 70, 333)
    assert (got - ref).abs().max().item() <= 1e-6      # |x|^2 ~ 0.37: a few ulp of the expansion terms


def test_index_points(cuda_device):
    from catgrasp_amd import pointnet2 as p2
    rng = np.random.default_rng(3)
    pts = torch.from_numpy(rng.normal(size=(3, 50, 7)).astype(np.float32))
    for shape in [(3, 11), (3, 5, 4)]:
        idx = torch.from_numpy(rng.integers(0, 50, shape))
        got = p2.index_points(pts.to(cuda_device), idx.to(cuda_device)).cpu()
        assert torch.equal(got, oref.index_points(pts, idx))
    with pytest.raises(IndexError):
        p2.index_points(pts.to(cuda_device), torch.full((3, 2), 50, dtype=torch.long, device=cuda_device))


@pytest.mark.parametrize('B,N,npoint', [(2, 1000, 64), (1, 5000, 128), (1, 20000, 256), (1, 30000, 40), (3, 64, 64)])
def test_farthest_point_sample_exact(cuda_device, B, N, npoint):
    from catgrasp_amd import pointnet2 as p2
    xyz = _cloud(B, N, 4 + N)
    start = torch.from_numpy(np.random.default_rng(5).integers(0, N, B))
    got = p2.farthest_point_sample(xyz.to(cuda_device), npoint, start=start).cpu()
    ref = oref.farthest_point_sample(xyz, npoint, start)
    assert got.dtype == torch.int64 and got.shape == (B, npoint)
    assert torch.equal(got, ref)


def test_farthest_point_sample_default_start_follows_torch_seed(cuda_device):
    """pointnet2.py:66 draws the start on the CPU generator: same torch seed -> same samples as the reference."""
    from catgrasp_amd import pointnet2 as p2
    xyz = _cloud(2, 800, 9)
    torch.manual_seed(123)
    got = p2.farthest_point_sample(xyz.to(cuda_device), 32).cpu()
    torch.manual_seed(123)
    start = torch.randint(0, 800, (2,), dtype=torch.long)
    assert torch.equal(got, oref.farthest_point_sample(xyz, 32, start))


@pytest.mark.parametrize('N,S,nsample,radius', [(2000, 100, 32, 0.02), (20000, 256, 32, 0.05), (500, 40, 16, 0.001), (300, 20, 64, 10.0)])
def test_query_ball_point(cuda_device, N, S, nsample, radius):
    from catgrasp_amd import pointnet2 as p2
    xyz = _cloud(2, N, 11)
    new_xyz = xyz[:, :S].clone()
    new_xyz[:, -1] += 5.0      # a query with an empty ball -> sentinel N in every slot
    got = p2.query_ball_point(radius, nsample, xyz.to(cuda_device), new_xyz.to(cuda_device)).cpu()
    d = oref.square_distance(new_xyz, xyz)
    ref = oref.query_ball_point(radius, nsample, xyz, new_xyz, sqrdists=d)
    assert got.shape == ref.shape and got.dtype == torch.int64
    if radius < 1.0:
        assert (got[:, -1] == N).all() and (ref[:, -1] == N).all()
    same = (got == ref).all(dim=-1)
    # rows that differ must contain a point inside the rounding band of r^2
    r2 = np.float32(radius ** 2)
    band = ((d - r2).abs() <= 1e-6 * max(1.0, float(r2)) + 4e-7).any(dim=-1)
    assert bool((same | band).all()), f'{(~(same | band)).sum().item()} rows differ outside the rounding band'
    assert same.float().mean().item() > 0.95


def test_sample_and_group(cuda_device):
    from catgrasp_amd import pointnet2 as p2
    B, N, D = 2, 3000, 5
    xyz = _cloud(B, N, 21)
    feats = torch.from_numpy(np.random.default_rng(22).normal(size=(B, N, D)).astype(np.float32))
    start = torch.tensor([7, 99])
    new_xyz, new_points, grouped_xyz, fps_idx = p2.sample_and_group(64, 0.03, 16, xyz.to(cuda_device), feats.to(cuda_device),
                                                                    returnfps=True, start=start)
    r_new_xyz, r_new_points, r_grouped, r_fps = oref.sample_and_group(64, 0.03, 16, xyz, feats, start)
    assert torch.equal(fps_idx.cpu(), r_fps)
    assert torch.equal(new_xyz.cpu(), r_new_xyz)
    rows_same = (grouped_xyz.cpu() == r_grouped).all(dim=-1).all(dim=-1)
    assert rows_same.float().mean().item() > 0.95
    m = rows_same
    assert torch.equal(new_points.cpu()[m], r_new_points[m])
    a, b = p2.sample_and_group(64, 0.03, 16, xyz.to(cuda_device), None, start=start)
    assert b.shape == (B, 64, 16, 3)
    nx, npnts = p2.sample_and_group_all(xyz.to(cuda_device), feats.to(cuda_device))
    rx, rp = oref.sample_and_group_all(xyz, feats)
    assert torch.equal(nx.cpu(), rx) and torch.equal(npnts.cpu(), rp)
