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


@pytest.fixture(params=['default', 'plain'])
def fps_kernel(request, monkeypatch):
    """Clouds of 2,049 .. 24,576 points are sampled by the blob-skipping kernel; CATGRASP_AMD_FPS=plain selects the round that updates
    every point.  Both must reproduce the oracle."""
    if request.param == 'plain':
        monkeypatch.setenv('CATGRASP_AMD_FPS', 'plain')
    else:
        monkeypatch.delenv('CATGRASP_AMD_FPS', raising=False)
    return request.param


@pytest.mark.parametrize('B,N,npoint', [(2, 1000, 64), (1, 2049, 100), (2, 4096, 96), (1, 5000, 128), (1, 8192, 64), (2, 8193, 64), (1, 12288, 64),
                                        (1, 15000, 64), (1, 20000, 256), (2, 20480, 48), (1, 24576, 48), (1, 30000, 40), (3, 64, 64),
                                        (4, 512, 128), (2, 513, 100), (2, 1024, 1024), (1, 1, 1), (2, 65, 65), (3, 256, 256), (2, 257, 30), (2, 1025, 64)])
def test_farthest_point_sample_exact(cuda_device, fps_kernel, B, N, npoint):
    from catgrasp_amd import pointnet2 as p2
    xyz = _cloud(B, N, 4 + N)
    start = torch.from_numpy(np.random.default_rng(5).integers(0, N, B))
    got = p2.farthest_point_sample(xyz.to(cuda_device), npoint, start=start).cpu()
    ref = oref.farthest_point_sample(xyz, npoint, start)
    assert got.dtype == torch.int64 and got.shape == (B, npoint)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('N,npoint', [(700, 200), (512, 300), (1000, 400), (200, 200), (3000, 200), (6000, 300), (10000, 100), (20000, 200), (22000, 64)])
def test_farthest_point_sample_ties_take_the_first_index(cuda_device, fps_kernel, N, npoint):
    """Equal running distances -- duplicate points (a cloud resampled with replacement) and an integer lattice -- must resolve to the
    smallest point index like torch.max (pointnet2.py:74), in every kernel geometry, also when the tied points sit in different
    lanes, slots, groups and waves (the blob-skipping kernel holds the points in spatial order, not index order)."""
    from catgrasp_amd import pointnet2 as p2
    rng = np.random.default_rng(N)
    base = rng.normal(0, 0.05, (N // 7, 3)).astype(np.float32)
    dup = base[rng.integers(0, len(base), N)]                                    # every point ~7 times, scattered over the indices
    lat = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing='ij'), -1).reshape(-1, 3).astype(np.float32)
    lat = lat[rng.permutation(len(lat))[:N]] if N <= len(lat) else np.concatenate([lat, lat[rng.integers(0, len(lat), N - len(lat))]])
    xyz = torch.from_numpy(np.stack([dup, lat * 0.01]))
    start = torch.tensor([3, N - 1])
    got = p2.farthest_point_sample(xyz.to(cuda_device), npoint, start=start).cpu()
    assert torch.equal(got, oref.farthest_point_sample(xyz, npoint, start))


def test_farthest_point_sample_degenerate_clouds(cuda_device, fps_kernel):
    """One point repeated (every running distance 0 after the first round: the padding slots of the kernel tie with the real points and
    must lose), the same with a single outlier, and a cloud with more samples asked for than it has distinct points."""
    from catgrasp_amd import pointnet2 as p2
    same = np.full((9000, 3), 0.25, np.float32)
    outlier = same.copy(); outlier[8999] = (1.0, 2.0, 3.0)
    few = np.random.default_rng(0).normal(size=(5, 3)).astype(np.float32)[np.random.default_rng(1).integers(0, 5, 9000)]
    xyz = torch.from_numpy(np.stack([same, outlier, few]))
    start = torch.tensor([5, 5, 7])
    got = p2.farthest_point_sample(xyz.to(cuda_device), 24, start=start).cpu()
    assert torch.equal(got, oref.farthest_point_sample(xyz, 24, start))


def test_farthest_point_sample_degenerate_small_clouds(cuda_device, fps_kernel):
    """The one-wavefront kernel of clouds up to 512 points (and its neighbours in size) on the same degenerate inputs: one repeated point (the padding slots tie with
    the real points at distance 0 and must lose), a single outlier, fewer distinct points than samples."""
    from catgrasp_amd import pointnet2 as p2
    for n in (70, 300, 512, 900):
        same = np.full((n, 3), -0.5, np.float32)
        outlier = same.copy(); outlier[n - 1] = (1.0, 2.0, 3.0)
        few = np.random.default_rng(0).normal(size=(3, 3)).astype(np.float32)[np.random.default_rng(1).integers(0, 3, n)]
        xyz = torch.from_numpy(np.stack([same, outlier, few]))
        start = torch.tensor([n - 1, 0, n // 2])
        got, got_xyz = p2.farthest_point_sample(xyz.to(cuda_device), 40, start=start, return_xyz=True)
        ref = oref.farthest_point_sample(xyz, 40, start)
        assert torch.equal(got.cpu(), ref), n
        assert torch.equal(got_xyz.cpu(), oref.index_points(xyz, ref)), n


def test_farthest_point_sample_skips_nothing_it_should_not(cuda_device, fps_kernel):
    """A cloud made of far-apart tight clusters plus a few stragglers: most blobs are skipped in most rounds, and the box test must stay
    on the safe side at cluster borders (the bound is evaluated with the update's own float operations)."""
    from catgrasp_amd import pointnet2 as p2
    rng = np.random.default_rng(11)
    centres = rng.uniform(-1, 1, (12, 3))
    pts = (centres[rng.integers(0, 12, 18000)] + rng.normal(0, 1e-3, (18000, 3))).astype(np.float32)
    pts[rng.integers(0, 18000, 40)] = rng.uniform(-1.5, 1.5, (40, 3)).astype(np.float32)
    xyz = torch.from_numpy(pts[None])
    start = torch.tensor([17])
    got = p2.farthest_point_sample(xyz.to(cuda_device), 400, start=start).cpu()
    assert torch.equal(got, oref.farthest_point_sample(xyz, 400, start))


@pytest.mark.parametrize('B,N,npoint', [(2, 900, 50), (1, 5000, 64), (2, 20000, 64), (1, 30000, 20)])
def test_farthest_point_sample_also_returns_the_sampled_points(cuda_device, fps_kernel, B, N, npoint):
    """cg_farthest_point_sample_xyz: the same samples, plus new_xyz == index_points(xyz, fps_idx) bit for bit (pointnet2.py:110-112), from
    every kernel geometry."""
    from catgrasp_amd import pointnet2 as p2
    xyz = _cloud(B, N, 31 + N).to(cuda_device)
    start = torch.from_numpy(np.random.default_rng(6).integers(0, N, B))
    idx, new_xyz = p2.farthest_point_sample(xyz, npoint, start=start, return_xyz=True)
    assert torch.equal(idx, p2.farthest_point_sample(xyz, npoint, start=start))
    assert new_xyz.shape == (B, npoint, 3) and torch.equal(new_xyz, p2.index_points(xyz, idx))


def test_farthest_point_sample_matches_the_reference_at_working_sizes(cuda_device, fps_kernel):
    """tests/golden/fps_large_golden.npz holds the REAL reference's samples (pointnet2.py:54-75, imported by
    tests/golden/make_golden_fps_large.py) for 3,000 .. 24,000-point clouds -- filled volume, surface, duplicated points, lattice:
    both kernels against the reference itself, at the sizes and in the tie situations they are built for."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fps_clouds
    from catgrasp_amd import pointnet2 as p2
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fps_large_golden.npz'))
    for kind, n, seed in fps_clouds.CASES:
        xyz = torch.from_numpy(fps_clouds.cloud(kind, n, seed)[None]).to(cuda_device)
        torch.manual_seed(seed)
        got = p2.farthest_point_sample(xyz, fps_clouds.NPOINT).cpu().numpy()          # the start is drawn like pointnet2.py:66 does
        assert np.array_equal(got, gold[f'{kind}_{n}_fps']), (kind, n)


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


# register-resident kernel: 1..3 layers of width 32 / 64 / 128, one / two / four neighbourhoods per 32-row tile (K <= 32 / 16 / 8) and
# several row tiles (K = 40, 64); strip kernel: a 256-wide layer, a 96-wide layer, four layers
@pytest.mark.parametrize('D,K,mlp', [(6, 32, [64, 64, 128]), (0, 16, [32, 64]), (13, 64, [128, 128, 256]), (3, 40, [64]), (6, 8, [32, 32, 64]),
                                     (2, 12, [128, 128, 128]), (6, 5, [128]), (0, 32, [32]), (9, 16, [64, 128]), (6, 33, [128, 64, 32]),
                                     (6, 32, [64, 96, 128]), (3, 16, [32, 32, 64, 64]), (6, 24, [128, 128, 128])])
def test_fused_set_abstraction_matches_sample_and_group_plus_torch_ops(cuda_device, D, K, mlp):
    """Row X1: group -> shared MLP -> max in one kernel vs the reference pipeline it replaces: sample_and_group
    (pointnet2.py:101-129, restated in oracle/pointnet_ref.py and pinned to the real one) followed by the torch
    Conv2d / BatchNorm2d(eval) / ReLU / max ops, float32, on the same FPS start and the same (exact) ball-query indices."""
    import torch.nn.functional as F
    from catgrasp_amd import pointnet2 as p2
    from oracle import pointnet_ref as oref
    torch.manual_seed(7)
    B, N, S = 3, 1500, 96
    xyz = torch.rand(B, N, 3) * 0.4
    pts = torch.randn(B, N, D) * 0.5 if D else None
    sa = p2.PointNetSetAbstraction(S, 0.08, K, 3 + D, mlp)
    with torch.no_grad():
        for bn in sa.mlp_bns:
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
    start = torch.tensor([5, 700, 1499])
    new_xyz_r, new_points_r, _, fps_r = oref.sample_and_group(S, 0.08, K, xyz, pts, start=start)
    # ball-query membership can legitimately flip inside the float rounding band of d^2 around r^2 (tested in
    # test_query_ball_point); group the oracle on the device's neighbour lists so this test isolates the fused layer
    idx = p2.query_ball_point(0.08, K, xyz.cuda(), new_xyz_r.cuda()).cpu()
    assert (idx != oref.query_ball_point(0.08, K, xyz, new_xyz_r)).float().mean().item() < 1e-3
    grouped = oref.index_points(xyz, idx) - new_xyz_r.view(B, S, 1, 3)
    new_points_r = torch.cat([grouped, oref.index_points(pts, idx)], dim=-1) if D else grouped
    h = new_points_r.permute(0, 3, 2, 1)
    sa.eval()
    with torch.no_grad():
        for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
            h = F.relu(bn(conv(h)))
        ref = torch.max(h, 2)[0].permute(0, 2, 1)
        sa.cuda()
        new_xyz, new_points = sa(xyz.cuda(), pts.cuda() if D else None, start=start)
    assert torch.equal(new_xyz.cpu(), new_xyz_r)
    assert new_points.shape == ref.shape == (B, S, mlp[-1])
    assert (new_points.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # grad-enabled call: the differentiable torch path over the grouped tensor computes the same function
    with torch.enable_grad():
        _, np2 = sa(xyz.cuda(), pts.cuda() if D else None, start=start)
    assert (np2.detach().cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
