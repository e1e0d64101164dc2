"""Row X1: the PointNet++ set-abstraction ENCODER past its first layer -- the LDS-tile fused layer (csrc/sa_tile.hip:
wide inputs, wide layers, any K), the group-all layer (GEMM chain and fused), the multi-scale layer and the 3-level stack -- against
the oracle: the restated primitives of pointnet2.py:54-149 (pinned to the imported reference) + torch float32 Conv2d / BN / ReLU / max."""
import numpy as np
import pytest
import torch

from oracle import pointnet_ref as oref
from oracle import setabstraction_ref as sref

pytestmark = pytest.mark.gpu


def _randomize_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
                m.weight.uniform_(0.5, 1.5, generator=g); m.bias.normal_(0, 0.1, generator=g)


def _relerr(a, b):
    """max over the ELEMENTS of |a - b| / max(1, |b|) -- the suite's standard (tests/test_pointnet_gpu.py::_close): 1e-4 absolute on
    O(1) values, 1e-4 relative on large ones, every element on its own scale."""
    return ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()


def _assert_ball_lists_differ_only_inside_the_rounding_band(got, want, xyz, new_xyz, radius, K):
    """Device neighbour lists `got` (B,S,K) against the oracle's / the reference's `want`: every row (centroid) whose list differs must be
    EXPLAINED by points whose squared distance lies inside the float rounding band of r^2 (exact outside
    |d^2 - r^2| < 1e-6; the band of tests/test_primitives_gpu.py::test_query_ball_point).  Explained = the device row is the reference's
    rule (first K in-radius indices in ascending order, padded with the first hit; pointnet2.py:78-98) applied to a membership that
    agrees with the float32 expansion everywhere outside the band.  -> number of differing rows."""
    B, S, _ = got.shape
    N = xyz.shape[1]
    r2 = float(np.float32(radius ** 2))
    tol = 1e-6 * max(1.0, r2) + 4e-7
    n_diff = 0
    for b in range(B):
        rows = (got[b] != want[b]).any(-1).nonzero()[:, 0].tolist()
        if not rows:
            continue
        d = oref.square_distance(new_xyz[b:b + 1, rows], xyz[b:b + 1])[0]            # (rows, N) float32, the reference's expansion
        for j, s in enumerate(rows):
            n_diff += 1
            band = (d[j] - r2).abs() <= tol
            sure = (d[j] <= r2) & ~band
            assert bool(band.any()), f'cloud {b} centroid {s}: lists differ but no point lies inside the rounding band'
            L = got[b, s]
            members = torch.unique(L)                                             # ascending
            m = len(members)
            assert bool((members < N).all()) and bool((sure | band)[members].all()), f'cloud {b} centroid {s}: a listed point is outside the radius'
            assert torch.equal(L[:m], members) and bool((L[m:] == L[0]).all()), f'cloud {b} centroid {s}: not ascending + padded with the first hit'
            horizon = int(members[-1]) if m == K else N - 1                       # a full list stops at its last member
            must = sure.clone(); must[horizon + 1:] = False
            listed = torch.zeros(N, dtype=torch.bool); listed[members] = True
            assert bool(listed[must].all()), f'cloud {b} centroid {s}: an in-radius point outside the band was skipped'
    return n_diff


# D (features), K, mlp, kind.  SA2 of the SSG stack; the MSG scales (3 + 320 inputs, K = 128 = two row tiles, a 96-wide layer); a hidden
# layer of 512 (the WIDE instance); 8 / 4 / 2 neighbourhoods per tile; D not a multiple of 4 (scalar gather); no features at all; one layer;
# a width that leaves waves without a block; first-layer shapes forced onto the tile kernel
@pytest.mark.parametrize('D,K,mlp,kind', [(128, 64, [128, 128, 256], None), (320, 128, [128, 128, 256], None), (64, 16, [64, 96, 128], None),
                                          (32, 32, [256, 512, 1024], None), (20, 8, [64, 64], None), (13, 24, [128, 128, 512], None),
                                          (0, 32, [64, 64, 128], 'tile'), (6, 100, [32], 'tile'), (128, 5, [160, 288], None),
                                          (61, 33, [32, 64, 96, 128], None), (3, 16, [128, 128, 256], 'tile'), (512, 48, [384, 256], None)])
def test_tile_set_abstraction_matches_grouping_plus_torch_ops(cuda_device, D, K, mlp, kind):
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    torch.manual_seed(11)
    B, N, S, r = 3, 700, 70, 0.12
    xyz = torch.rand(B, N, 3) * 0.5
    pts = torch.randn(B, N, D) * 0.5 if D else None
    sa = p2.PointNetSetAbstraction(S, r, K, 3 + D, mlp)
    _randomize_bn(sa, 5)
    sa.eval()
    start = torch.tensor([3, 300, 699])
    fps = oref.farthest_point_sample(xyz, S, start)
    new_xyz = oref.index_points(xyz, fps)
    idx = p2.query_ball_point(r, K, xyz.cuda(), new_xyz.cuda()).cpu()
    _assert_ball_lists_differ_only_inside_the_rounding_band(idx, oref.query_ball_point(r, K, xyz, new_xyz), xyz, new_xyz, r, K)
    layers = sref.layers_of(sa.state_dict(), '', len(mlp))
    _, ref, _, _ = sref.sa_forward(xyz, pts, S, r, K, layers, start, idx=idx)
    W = prim.SetAbstractionWeights([(w.double().numpy(), b.double().numpy(), tuple(t.double().numpy() for t in (g, be, mu, var)))
                                    for w, b, g, be, mu, var in layers], 3 + D, cuda_device, kind=kind)
    assert W.kind == 'tile'
    got_cs = prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), idx.cuda(), W)                       # (B, C, S)
    got_cl = prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), idx.cuda(), W, channels_last=True)   # (B, S, C)
    assert got_cs.shape == (B, mlp[-1], S) and got_cl.shape == (B, S, mlp[-1])
    assert torch.equal(got_cs.permute(0, 2, 1), got_cl)
    assert _relerr(got_cl.cpu(), ref) <= 1e-5
    # a channel slice of a wider output (the multi-scale layer's concatenation): neighbours untouched
    wide = torch.full((B, S, mlp[-1] + 64), -7.0, device=cuda_device)
    prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), idx.cuda(), W, channels_last=True, out=wide[:, :, 32:32 + mlp[-1]])
    assert torch.equal(wide[:, :, 32:32 + mlp[-1]], got_cl) and bool((wide[:, :, :32] == -7).all()) and bool((wide[:, :, 32 + mlp[-1]:] == -7).all())
    # an out-of-range index is reported like index_points does
    bad = idx.clone(); bad[1, 5, 0] = N
    with pytest.raises(IndexError):
        prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), bad.cuda(), W)
    if kind is None:
        # through the module (FPS + ball query + the fused kernel), eval and grad-enabled
        sa.cuda()
        with torch.no_grad():
            nx, np_ = sa(xyz.cuda(), pts.cuda() if D else None, start=start)
        assert torch.equal(nx.cpu(), new_xyz) and _relerr(np_.cpu(), ref) <= 1e-5
        with torch.enable_grad():
            _, np2 = sa(xyz.cuda(), pts.cuda() if D else None, start=start)
        assert _relerr(np2.detach().cpu(), ref) <= 1e-4


def test_first_layer_reg_kernel_strided_output(cuda_device):
    """The register-resident first-layer kernel writing (B,S,C) rows / a channel slice (cg_sa_group_mlp_max_strided)."""
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    torch.manual_seed(2)
    B, N, S, K, D = 2, 900, 50, 32, 3
    xyz = torch.rand(B, N, 3) * 0.4; pts = torch.randn(B, N, D)
    sa = p2.PointNetSetAbstraction(S, 0.1, K, 3 + D, [64, 64, 128]); _randomize_bn(sa, 1); sa.eval()
    start = torch.tensor([0, 1])
    layers = sref.layers_of(sa.state_dict(), '', 3)
    new_xyz = oref.index_points(xyz, oref.farthest_point_sample(xyz, S, start))
    idx = p2.query_ball_point(0.1, K, xyz.cuda(), new_xyz.cuda())
    _, ref, _, _ = sref.sa_forward(xyz, pts, S, 0.1, K, layers, start, idx=idx.cpu())
    W = sa.cuda()._weights(cuda_device)
    assert W.kind == 'reg'
    a = prim.group_mlp_max(xyz.cuda(), pts.cuda(), new_xyz.cuda(), idx, W)
    b = prim.group_mlp_max(xyz.cuda(), pts.cuda(), new_xyz.cuda(), idx, W, channels_last=True)
    assert torch.equal(a.permute(0, 2, 1), b) and _relerr(b.cpu(), ref) <= 1e-5


@pytest.mark.parametrize('B,N,D,mlp', [(1, 128, 256, [256, 512, 1024]), (8, 128, 256, [256, 512, 1024]), (3, 200, 640, [256, 512, 1024]),
                                        (2, 77, 5, [64, 128]), (2, 1000, 0, [64, 128, 1024]), (1, 130, 61, [32])])
def test_group_all_layer_gemm_chain_and_fused(cuda_device, B, N, D, mlp):
    """sample_and_group_all (pointnet2.py:132-149) + shared MLP + max over all points: both execution plans against the torch ops."""
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    torch.manual_seed(4)
    xyz = torch.rand(B, N, 3) - 0.5
    pts = torch.randn(B, N, D) * 0.5 if D else None
    sa = p2.PointNetSetAbstraction(None, None, None, 3 + D, mlp, group_all=True); _randomize_bn(sa, 9); sa.eval()
    layers = sref.layers_of(sa.state_dict(), '', len(mlp))
    ref = sref.sa_all_forward(xyz, pts, layers)
    sa.cuda()
    W = sa._weights(cuda_device)
    chain = prim.group_all_mlp_max(xyz.cuda(), pts.cuda() if D else None, W, fused=False)
    assert chain.shape == (B, mlp[-1]) and _relerr(chain.cpu(), ref) <= 1e-5
    if W.cin[0] <= W.TILE_MAX_CIN:
        fused = prim.group_all_mlp_max(xyz.cuda(), pts.cuda() if D else None, W, fused=True)
        assert _relerr(fused.cpu(), ref) <= 1e-5
    with torch.no_grad():
        nx, out = sa(xyz.cuda(), pts.cuda() if D else None)
    assert nx.shape == (B, 1, 3) and not bool(nx.any()) and out.shape == (B, 1, mlp[-1]) and _relerr(out[:, 0].cpu(), ref) <= 1e-5


def _scene_cloud(B, N, seed):
    """20k-point 'clutter': points on a handful of small boxes / cylinders inside the unit ball, with unit normals as features."""
    rng = np.random.default_rng(seed)
    out = np.zeros((B, N, 6), np.float32)
    for b in range(B):
        k = 8
        centres = rng.uniform(-0.55, 0.55, size=(k, 3))
        which = rng.integers(0, k, size=N)
        local = rng.normal(size=(N, 3)); local /= np.linalg.norm(local, axis=1, keepdims=True)
        radius = rng.uniform(0.08, 0.25, size=k)[which]
        out[b, :, :3] = centres[which] + local * radius[:, None] * rng.uniform(0.9, 1.0, size=(N, 1))
        out[b, :, 3:] = local
    return torch.from_numpy(out)


@pytest.mark.parametrize('B', [1, 8])
def test_pointnet2_encoder_ssg_20k_points(cuda_device, B):
    """The 3-level single-scale stack (512 / 0.2 / 32 -> 128 / 0.4 / 64 -> all) on 20,000-point clouds: every level against the oracle
    (sample indices exact; neighbour lists equal outside the rounding band; features within 1e-4)."""
    from catgrasp_amd import pointnet2 as p2
    N = 20000
    x = _scene_cloud(B, N, 31 + B)
    enc = p2.PointNet2Encoder(channel=6); _randomize_bn(enc, 3); enc.eval()
    sd = enc.state_dict()
    torch.manual_seed(5)
    start = (torch.randint(0, N, (B,)), torch.randint(0, 512, (B,)))
    enc.cuda()
    xd = x.to(cuda_device)
    with torch.no_grad():
        g, ((l1_xyz, l1_pts), (l2_xyz, l2_pts)) = enc(xd, start=start)
    xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
    # level 1
    fps1 = oref.farthest_point_sample(xyz, 512, start[0])
    nx1 = oref.index_points(xyz, fps1)
    assert torch.equal(l1_xyz.cpu(), nx1)
    idx1 = p2.query_ball_point(0.2, 32, xd[:, :, :3].contiguous(), l1_xyz).cpu()
    for b in range(B):          # cloud by cloud: the oracle's (S, N) int64 sort of one cloud at a time
        _assert_ball_lists_differ_only_inside_the_rounding_band(idx1[b:b + 1], oref.query_ball_point(0.2, 32, xyz[b:b + 1], nx1[b:b + 1]),
                                                                xyz[b:b + 1], nx1[b:b + 1], 0.2, 32)
    _, r1, _, _ = sref.sa_forward(xyz, feats, 512, 0.2, 32, sref.layers_of(sd, 'sa1.', 3), start[0], idx=idx1)
    assert _relerr(l1_pts.cpu(), r1) <= 1e-4
    # level 2 (on the ORACLE's level-1 output: errors may accumulate through the stack, the bar stays 1e-4)
    fps2 = oref.farthest_point_sample(nx1, 128, start[1])
    nx2 = oref.index_points(nx1, fps2)
    assert torch.equal(l2_xyz.cpu(), nx2)
    idx2 = p2.query_ball_point(0.4, 64, l1_xyz, l2_xyz).cpu()
    _assert_ball_lists_differ_only_inside_the_rounding_band(idx2, oref.query_ball_point(0.4, 64, nx1, nx2), nx1, nx2, 0.4, 64)
    _, r2, _, _ = sref.sa_forward(nx1, r1, 128, 0.4, 64, sref.layers_of(sd, 'sa2.', 3), start[1], idx=idx2)
    assert _relerr(l2_pts.cpu(), r2) <= 1e-4
    # level 3
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 3))
    assert g.shape == (B, 1024) and _relerr(g.cpu(), r3) <= 1e-4
    # float64 evaluation of the same stack: the kernels are as close to it as the float32 oracle is
    _, t1, _, _ = sref.sa_forward(xyz.double(), feats.double(), 512, 0.2, 32, sref.layers_of(sd, 'sa1.', 3), start[0], idx=idx1, dtype=torch.float64)
    _, t2, _, _ = sref.sa_forward(nx1.double(), t1, 128, 0.4, 64, sref.layers_of(sd, 'sa2.', 3), start[1], idx=idx2, dtype=torch.float64)
    t3 = sref.sa_all_forward(nx2.double(), t2, sref.layers_of(sd, 'sa3.', 3), dtype=torch.float64)
    assert _relerr(g.cpu().double(), t3) <= max(2 * _relerr(r3.double(), t3), 2e-6)


def test_pointnet2_encoder_msg(cuda_device):
    """The multi-scale stack (3 radii per level, 3 + 320 / 3 + 640 inputs further up) on a 20,000-point cloud."""
    from catgrasp_amd import pointnet2 as p2
    B, N = 2, 20000
    x = _scene_cloud(B, N, 77)
    enc = p2.PointNet2Encoder(channel=6, msg=True); _randomize_bn(enc, 8); enc.eval()
    sd = enc.state_dict()
    start = (torch.tensor([17, 19999]), torch.tensor([0, 511]))
    enc.cuda()
    xd = x.to(cuda_device)
    with torch.no_grad():
        g, ((l1_xyz, l1_pts), (l2_xyz, l2_pts)) = enc(xd, start=start)
    xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()

    def msg_layers(prefix, n_scales):
        return [sref.layers_of(sd, prefix, 3, conv=f'conv_blocks.{i}', bn=f'bn_blocks.{i}') for i in range(n_scales)]
    radii1, ks1 = (0.1, 0.2, 0.4), (16, 32, 128)
    nx1 = oref.index_points(xyz, oref.farthest_point_sample(xyz, 512, start[0]))
    assert torch.equal(l1_xyz.cpu(), nx1)
    idx1 = [p2.query_ball_point(r, k, xd[:, :, :3].contiguous(), l1_xyz).cpu() for r, k in zip(radii1, ks1)]
    for i1, r, k in zip(idx1, radii1, ks1):
        _assert_ball_lists_differ_only_inside_the_rounding_band(i1, oref.query_ball_point(r, k, xyz, nx1), xyz, nx1, r, k)
    _, r1, _, _ = sref.sa_msg_forward(xyz, feats, 512, radii1, ks1, msg_layers('sa1.', 3), start[0], idx_list=idx1)
    assert l1_pts.shape == (B, 512, 320) and _relerr(l1_pts.cpu(), r1) <= 1e-4
    radii2, ks2 = (0.2, 0.4, 0.8), (32, 64, 128)
    nx2 = oref.index_points(nx1, oref.farthest_point_sample(nx1, 128, start[1]))
    assert torch.equal(l2_xyz.cpu(), nx2)
    idx2 = [p2.query_ball_point(r, k, l1_xyz, l2_xyz).cpu() for r, k in zip(radii2, ks2)]
    for i2, r, k in zip(idx2, radii2, ks2):
        _assert_ball_lists_differ_only_inside_the_rounding_band(i2, oref.query_ball_point(r, k, nx1, nx2), nx1, nx2, r, k)
    _, r2, _, _ = sref.sa_msg_forward(nx1, r1, 128, radii2, ks2, msg_layers('sa2.', 3), start[1], idx_list=idx2)
    assert l2_pts.shape == (B, 128, 640) and _relerr(l2_pts.cpu(), r2) <= 1e-4
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 3))
    assert _relerr(g.cpu(), r3) <= 1e-4


def test_encoder_state_dict_train_eval_roundtrip(cuda_device):
    """Weights changed after a forward are picked up (cache keyed on tensor versions); train-mode output == eval-mode HIP output when the
    BatchNorm statistics are frozen equal (momentum 0), i.e. the torch path and the kernels compute the same function."""
    from catgrasp_amd import pointnet2 as p2
    torch.manual_seed(0)
    enc = p2.PointNet2Encoder(channel=3, npoints=(64, 16), radii=(0.3, 0.6), nsamples=(8, 16)); _randomize_bn(enc, 2); enc.cuda()
    x = torch.rand(2, 500, 3, device=cuda_device)
    start = (torch.tensor([1, 2]), torch.tensor([3, 4]))
    enc.eval()
    with torch.no_grad():
        g0, _ = enc(x, start=start)
    with torch.enable_grad():
        g1, _ = enc(x, start=start)              # eval-mode BN, torch ops
    assert _relerr(g0.cpu(), g1.detach().cpu()) <= 1e-4
    with torch.no_grad():
        enc.sa3.mlp_convs[2].bias.add_(1.0)
        g2, _ = enc(x, start=start)
    assert (g2 - g0).abs().max().item() > 0.1
    with pytest.raises(RuntimeError), torch.no_grad():
        enc(x.cpu(), start=start)                # eval mode, no grad, CPU tensor: there is no CPU inference path


def test_group_all_gemm_epilogue_max_and_appended_rows(cuda_device):
    """cg_gemm_bias_relu_groupmax (the group-all level's last layer with its max over the points in the GEMM epilogue): groups that are
    not a multiple of the 32-row tile (per-element atomics across a group boundary), many groups, a width that is not a multiple of 32;
    and the [features | xyz | pad] rows the tile kernel appends (append_xyz) == what cg_sa_concat_input builds."""
    import ctypes
    from catgrasp_amd import _lib as L
    from catgrasp_amd import folding, ops
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    g = torch.Generator().manual_seed(1)
    for groups, rows, K, N in ((1, 128, 264, 1024), (5, 77, 64, 96), (300, 20, 32, 40), (3, 1000, 128, 512)):
        x = torch.randn(groups * rows, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.2
        b = torch.randn(N, generator=g)
        ref = torch.relu(x @ w.t() + b).view(groups, rows, N).max(dim=1)[0]
        wp = torch.from_numpy(folding.pack_b(w.numpy())).to(cuda_device)
        xd, bd = x.to(cuda_device), b.to(cuda_device)
        out = torch.full((groups, N), -5.0, device=cuda_device)
        L.check(L.lib().cg_gemm_bias_relu_groupmax(L._p(xd), ctypes.c_int(groups * rows), ctypes.c_int(K), ctypes.c_int(K), L._p(wp), ctypes.c_int(N),
                                                   L._p(bd), ctypes.c_int(rows), L._p(out), L._stream()), 'cg_gemm_bias_relu_groupmax')
        two_step = ops.group_max(ops.gemm_bias_act(xd, wp, N, bias=bd, relu=True), groups)
        assert torch.equal(out, two_step)                     # same products, same order; the max is order-free
        assert _relerr(out.cpu(), ref) <= 1e-4
    # a NaN activation propagates to the pooled feature, as torch.max does (the integer atomic max sees 0x7fc00000 above every finite float)
    x = torch.randn(64, 32, generator=g); x[5, 3] = float('nan')
    w = torch.randn(32, 32, generator=g); b = torch.zeros(32)
    wp = torch.from_numpy(folding.pack_b(w.numpy())).to(cuda_device)
    xd, bd = x.to(cuda_device), b.to(cuda_device)
    out = torch.empty((2, 32), device=cuda_device)
    L.check(L.lib().cg_gemm_bias_relu_groupmax(L._p(xd), ctypes.c_int(64), ctypes.c_int(32), ctypes.c_int(32), L._p(wp), ctypes.c_int(32), L._p(bd),
                                               ctypes.c_int(32), L._p(out), L._stream()), 'cg_gemm_bias_relu_groupmax')
    assert bool(torch.isnan(out[0]).all()) and bool(torch.isfinite(out[1]).all())
    # appended rows
    torch.manual_seed(3)
    B, N, S, K, D = 2, 400, 37, 16, 20
    xyz = torch.rand(B, N, 3, device=cuda_device); pts = torch.randn(B, N, D, device=cuda_device)
    sa = p2.PointNetSetAbstraction(S, 0.3, K, 3 + D, [64, 160]); _randomize_bn(sa, 4); sa.eval(); sa.to(cuda_device)
    W = prim.SetAbstractionWeights(p2._sa_layers_from_state(sa.state_dict(), 'mlp_', 2), 3 + D, cuda_device, kind='tile')
    new_xyz = xyz[:, :S].contiguous()
    idx = p2.query_ball_point(0.3, K, xyz, new_xyz)
    plain = prim.group_mlp_max(xyz, pts, new_xyz, idx, W, channels_last=True)
    rows = torch.full((B, S, 168), 9.0, device=cuda_device)
    got = prim.group_mlp_max(xyz, pts, new_xyz, idx, W, channels_last=True, out=rows[:, :, :160], append_xyz=8)
    assert torch.equal(got, plain) and torch.equal(rows[:, :, :160], plain)
    want = torch.empty((B * S, 168), device=cuda_device)
    L.check(L.lib().cg_sa_concat_input(L._p(new_xyz), L._p(plain.contiguous()), ctypes.c_long(B * S), ctypes.c_int(160), ctypes.c_int(168), L._p(want),
                                       L._stream()), 'cg_sa_concat_input')
    assert torch.equal(rows.view(B * S, 168), want)


@pytest.mark.parametrize('B,N,S,K,D,mlp', [(20, 300, 128, 64, 16, [64, 96, 128]), (16, 600, 512, 16, 29, [128, 128, 256]), (8, 400, 128, 100, 8, [32, 256]),
                                             (4, 500, 128, 200, 128, [128, 128, 256]), (16, 256, 1024, 8, 0, [64, 64]), (20, 300, 128, 33, 64, [256, 128, 1024])])
def test_tile_kernel_with_128_row_tiles(cuda_device, B, N, S, K, D, mlp):
    """Launches large enough for the 128-row tile instance of sa_tile_kernel (>= 2 tiles per CU; hidden widths <= 256): one, two, eight
    and sixteen neighbourhoods per tile, two row tiles per neighbourhood (K = 200), a layer narrower than 128 (a wave takes half of the
    rows), against the torch ops on the gathered tensor (random neighbour lists: the ball query is not under test here)."""
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    g = torch.Generator().manual_seed(B * 1000 + K)
    xyz = torch.rand(B, N, 3, generator=g)
    pts = torch.randn(B, N, D, generator=g) * 0.5 if D else None
    new_xyz = xyz[:, torch.randint(0, N, (S,), generator=g)].contiguous()
    idx = torch.randint(0, N, (B, S, K), generator=g)
    sa = p2.PointNetSetAbstraction(S, 0.2, K, 3 + D, mlp); _randomize_bn(sa, 6); sa.eval()
    layers = sref.layers_of(sa.state_dict(), '', len(mlp))
    grouped = oref.index_points(xyz, idx) - new_xyz.view(B, S, 1, 3)
    if D:
        grouped = torch.cat([grouped, oref.index_points(pts, idx)], dim=-1)
    ref = sref.mlp_max(grouped, layers)
    W = prim.SetAbstractionWeights([(w.double().numpy(), b.double().numpy(), tuple(t.double().numpy() for t in (ga, be, mu, var)))
                                    for w, b, ga, be, mu, var in layers], 3 + D, cuda_device, kind='tile')
    got = prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), idx.cuda(), W, channels_last=True)
    assert _relerr(got.cpu(), ref) <= 1e-5
    got_cs = prim.group_mlp_max(xyz.cuda(), pts.cuda() if D else None, new_xyz.cuda(), idx.cuda(), W)
    assert torch.equal(got_cs.permute(0, 2, 1), got)


@pytest.mark.parametrize('D,K,mlp', [(13, 64, [128, 128, 256]), (6, 32, [64, 96, 128]), (3, 16, [32, 32, 64, 64]), (6, 40, [256])])
def test_first_level_shapes_on_both_kernel_families(cuda_device, D, K, mlp):
    """First-level shapes outside the register kernel's signatures go to the tile kernel by default (round 5: 1.3 - 1.9 x the LDS-strip
    kernel); the strip kernel of setabstraction.hip stays reachable (kind='reg') and both agree with the torch ops."""
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import primitives as prim
    torch.manual_seed(5)
    B, N, S = 2, 800, 60
    xyz = torch.rand(B, N, 3) * 0.4; pts = torch.randn(B, N, D) * 0.5
    sa = p2.PointNetSetAbstraction(S, 0.1, K, 3 + D, mlp); _randomize_bn(sa, 7); sa.eval()
    layers = sref.layers_of(sa.state_dict(), '', len(mlp))
    start = torch.tensor([1, 2])
    new_xyz = oref.index_points(xyz, oref.farthest_point_sample(xyz, S, start))
    idx = p2.query_ball_point(0.1, K, xyz.cuda(), new_xyz.cuda())
    _, ref, _, _ = sref.sa_forward(xyz, pts, S, 0.1, K, layers, start, idx=idx.cpu())
    packed = [(w.double().numpy(), b.double().numpy(), tuple(t.double().numpy() for t in (ga, be, mu, var))) for w, b, ga, be, mu, var in layers]
    auto = prim.SetAbstractionWeights(packed, 3 + D, cuda_device)
    assert auto.kind == ('reg' if max(mlp) <= 64 else 'tile')
    for kind in ('reg', 'tile'):
        W = prim.SetAbstractionWeights(packed, 3 + D, cuda_device, kind=kind)
        got = prim.group_mlp_max(xyz.cuda(), pts.cuda(), new_xyz.cuda(), idx, W, channels_last=True)
        assert _relerr(got.cpu(), ref) <= 1e-5, kind


def test_encoder_against_the_stack_built_from_the_real_reference_primitives(cuda_device):
    """The HIP encoder against tests/golden/pp_encoder_golden.npz: samples, neighbour lists and grouped tensors of that file come out of
    the REAL reference's sample_and_group / sample_and_group_all (tests/golden/make_golden_encoder.py), the MLPs out of torch.nn."""
    import test_pointnet2_encoder_cpu as cpu
    from catgrasp_amd import pointnet2 as p2
    g, sd, x, start = cpu._encoder_golden()
    enc = p2.PointNet2Encoder(**cpu.ENC_GOLDEN_CFG)
    enc.load_state_dict(sd); enc.eval().to(cuda_device)
    with torch.no_grad():
        gf, ((x1, p1), (x2, p2_)) = enc(x.to(cuda_device), start=start)
    assert torch.equal(x1.cpu(), torch.from_numpy(g['l1_xyz'])) and torch.equal(x2.cpu(), torch.from_numpy(g['l2_xyz']))
    assert _relerr(p1.cpu(), torch.from_numpy(g['l1_points'])) <= 1e-4
    assert _relerr(p2_.cpu(), torch.from_numpy(g['l2_points'])) <= 1e-4
    assert _relerr(gf.cpu(), torch.from_numpy(g['global_feat'])) <= 1e-4


def test_full_size_encoder_against_the_real_reference_primitives_at_20k_points(cuda_device):
    """the FULL-SIZE stack pinned to the reference itself.  tests/golden/pp_encoder_20k_golden.npz holds one
    20,000-point cloud pushed through the REAL reference's sample_and_group (512 / 0.2 / 32, then 128 / 0.4 / 64) and sample_and_group_all
    (tests/golden/make_golden_encoder.py::main_20k) with torch.nn Conv2d / BatchNorm2d / ReLU / max on the grouped tensors.  Samples
    exact; the device's neighbour lists against the reference's own lists (equal, or explained by the rounding band of r^2); features
    per element within 1e-4 * max(1, |reference|)."""
    import test_pointnet2_encoder_cpu as cpu
    from catgrasp_amd import pointnet2 as p2
    g, enc, x, start = cpu._encoder_golden_20k()
    enc.to(cuda_device)
    xd = x.to(cuda_device)
    with torch.no_grad():
        gf, ((x1, p1), (x2, p2_)) = enc(xd, start=start)
    xyz = x[:, :, :3].contiguous()
    fps1, fps2 = torch.from_numpy(g['fps1']).long(), torch.from_numpy(g['fps2']).long()
    nx1 = oref.index_points(xyz, fps1); nx2 = oref.index_points(nx1, fps2)
    assert torch.equal(x1.cpu(), nx1) and torch.equal(x2.cpu(), nx2)
    assert torch.equal(p2.farthest_point_sample(xd[:, :, :3].contiguous(), 512, start[0]).cpu(), fps1)
    idx1 = p2.query_ball_point(0.2, 32, xd[:, :, :3].contiguous(), x1).cpu()
    idx2 = p2.query_ball_point(0.4, 64, x1, x2).cpu()
    d1 = _assert_ball_lists_differ_only_inside_the_rounding_band(idx1, torch.from_numpy(g['idx1']).long(), xyz, nx1, 0.2, 32)
    d2 = _assert_ball_lists_differ_only_inside_the_rounding_band(idx2, torch.from_numpy(g['idx2']).long(), nx1, nx2, 0.4, 64)
    assert d1 <= 5 and d2 <= 2, (d1, d2)          # band rows are rare: a flipped list changes that centroid's features, see below
    ok1 = (idx1 == torch.from_numpy(g['idx1']).long()).all(-1)[0]
    ok2 = (idx2 == torch.from_numpy(g['idx2']).long()).all(-1)[0]
    # level 1: every centroid whose list equals the reference's.  A band flip changes that centroid's features, and through the level-2
    # grouping every level-2 row that lists it and the global feature: those are excluded, everything else is compared
    assert _relerr(p1.cpu()[0, ok1], torch.from_numpy(g['l1_points'])[0, ok1]) <= 1e-4
    clean2 = ok2 & ok1[idx2[0]].all(-1)
    assert int(clean2.sum()) >= 100
    assert _relerr(p2_.cpu()[0, clean2], torch.from_numpy(g['l2_points'])[0, clean2]) <= 1e-4
    if bool(clean2.all()):
        assert _relerr(gf.cpu(), torch.from_numpy(g['global_feat'])) <= 1e-4
