"""The differentiable torch path of the set-abstraction modules (training / grad-enabled calls; SURVEY §8 B2: the reference's trainers
must keep working) against the oracle stack, on CPU: same indices as the restated reference primitives, same features."""
import torch

from oracle import pointnet_ref as oref
from oracle import setabstraction_ref as sref


def test_torch_primitives_follow_the_reference():
    from catgrasp_amd import pointnet2 as p2
    torch.manual_seed(3)
    xyz = torch.rand(2, 400, 3)
    start = torch.tensor([5, 399])
    fps = p2._torch_fps(xyz, 40, start)
    assert torch.equal(fps, oref.farthest_point_sample(xyz, 40, start))
    new_xyz = oref.index_points(xyz, fps)
    assert torch.equal(p2._torch_index(xyz, fps), new_xyz)
    for r, k in ((0.15, 16), (0.05, 8), (5.0, 64)):
        assert torch.equal(p2._torch_ball(r, k, xyz, new_xyz), oref.query_ball_point(r, k, xyz, new_xyz))


def _bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
                m.weight.uniform_(0.5, 1.5, generator=g); m.bias.normal_(0, 0.1, generator=g)


def test_encoder_torch_path_equals_the_oracle_stack_and_is_differentiable():
    from catgrasp_amd import pointnet2 as p2
    torch.manual_seed(1)
    B, N = 2, 300
    x = torch.rand(B, N, 6)
    enc = p2.PointNet2Encoder(channel=6, npoints=(48, 12), radii=(0.3, 0.6), nsamples=(8, 16), mlps=((32, 32, 64), (64, 64, 128), (128, 256)))
    _bn(enc, 4); enc.eval()
    sd = enc.state_dict()
    assert len(sd) == 8 * 7            # 8 conv + bn pairs x (w, b, bn w, bn b, mean, var, num_batches_tracked)
    start = (torch.tensor([0, 7]), torch.tensor([3, 47]))
    xg = x.clone().requires_grad_(True)
    g, ((x1, p1), (x2, p2_)) = enc(xg, start=start)                 # eval-mode BatchNorm, grad enabled -> torch ops
    xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
    nx1, r1, _, _ = sref.sa_forward(xyz, feats, 48, 0.3, 8, sref.layers_of(sd, 'sa1.', 3), start[0])
    nx2, r2, _, _ = sref.sa_forward(nx1, r1, 12, 0.6, 16, sref.layers_of(sd, 'sa2.', 3), start[1])
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 2))
    assert torch.equal(x1, nx1) and torch.equal(x2, nx2)
    assert (p1 - r1).abs().max().item() <= 1e-5 and (p2_ - r2).abs().max().item() <= 1e-5 and (g - r3).abs().max().item() <= 1e-5
    g.sum().backward()
    assert xg.grad is not None and float(xg.grad.abs().sum()) > 0 and float(enc.sa1.mlp_convs[0].weight.grad.abs().sum()) > 0
    # multi-scale stack: shapes and the oracle
    enc = p2.PointNet2Encoder(channel=6, msg=True, npoints=(32, 8), radii=((0.2, 0.4), (0.4, 0.8)), nsamples=((4, 8), (8, 16)),
                              mlps=(((32, 32), (32, 64)), ((64, 64), (64, 96)), (128, 256)))
    _bn(enc, 6); enc.eval()
    sd = enc.state_dict()
    start = (torch.tensor([1, 2]), torch.tensor([3, 4]))
    with torch.enable_grad():
        g, ((x1, p1), (x2, p2_)) = enc(x, start=start)
    ml = lambda prefix, n: [sref.layers_of(sd, prefix, 2, conv=f'conv_blocks.{i}', bn=f'bn_blocks.{i}') for i in range(n)]
    nx1, r1, _, _ = sref.sa_msg_forward(xyz, feats, 32, (0.2, 0.4), (4, 8), ml('sa1.', 2), start[0])
    nx2, r2, _, _ = sref.sa_msg_forward(nx1, r1, 8, (0.4, 0.8), (8, 16), ml('sa2.', 2), start[1])
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 2))
    assert p1.shape == (B, 32, 96) and p2_.shape == (B, 8, 160)
    assert (p1 - r1).abs().max().item() <= 1e-5 and (p2_ - r2).abs().max().item() <= 1e-5 and (g - r3).abs().max().item() <= 1e-5


def test_set_abstraction_weights_layouts():
    """Host packing of the two kernel families: column order and padding of layer 0."""
    import numpy as np
    from catgrasp_amd import folding
    from catgrasp_amd.primitives import SetAbstractionWeights
    rng = np.random.default_rng(0)
    w0 = rng.normal(size=(64, 9)); b0 = rng.normal(size=64)
    reg = SetAbstractionWeights([(w0, b0, None)], 9, 'cpu')
    assert reg.kind == 'reg' and reg.cin == [16]
    assert np.array_equal(reg.w[0].numpy(), folding.pack_b(np.concatenate([w0, np.zeros((64, 7))], 1)))
    tile = SetAbstractionWeights([(w0, b0, None)], 9, 'cpu', kind='tile')
    assert tile.kind == 'tile' and tile.cin == [16]
    assert np.array_equal(tile.w[0].numpy(), folding.pack_b(np.concatenate([w0[:, 3:], w0[:, :3], np.zeros((64, 7))], 1)))
    w1 = rng.normal(size=(128, 131))
    t2 = SetAbstractionWeights([(w1, rng.normal(size=128), None), (rng.normal(size=(256, 128)), rng.normal(size=256), None)], 131, 'cpu')
    assert t2.kind == 'tile' and t2.cin == [136, 128] and t2.cout == [128, 256] and t2.hidden_max == 128


def _encoder_golden():
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pp_encoder_golden.npz'))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}
    x = torch.from_numpy(g['x'])
    B, N = x.shape[:2]
    starts = []
    for seed, n in zip(g['seeds'], (N, 96)):
        torch.manual_seed(int(seed))
        starts.append(torch.randint(0, n, (B,), dtype=torch.long))          # the reference's own draw, pointnet2.py:66
    return g, sd, x, tuple(starts)


def _encoder_golden_20k():
    """-> (golden file, the default-size encoder with the golden's seeded weights [sha256-pinned], x (1,20000,6), FPS starts)."""
    import hashlib
    import os
    import numpy as np
    from catgrasp_amd import pointnet2 as p2
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pp_encoder_20k_golden.npz'))
    torch.manual_seed(15)                                  # make_golden_encoder.py: WEIGHT_SEED_20K / BN_SEED_20K
    enc = p2.PointNet2Encoder(channel=6)
    _bn(enc, 16)
    enc.eval()
    h = hashlib.sha256()
    for k, v in enc.state_dict().items():
        h.update(k.encode()); h.update(v.numpy().tobytes())
    assert h.hexdigest() == str(g['weights_sha256']), 'the seeded weights of the 20k golden could not be regenerated bit for bit'
    x = torch.from_numpy(g['x'])
    starts = []
    for seed, n in zip(g['seeds'], (x.shape[1], 512)):
        torch.manual_seed(int(seed))
        starts.append(torch.randint(0, n, (1,), dtype=torch.long))          # the reference's own draw, pointnet2.py:66
    return g, enc, x, tuple(starts)


ENC_GOLDEN_CFG = dict(channel=6, npoints=(96, 24), radii=(0.25, 0.5), nsamples=(16, 32), mlps=((32, 64, 64), (64, 128, 128), (128, 256, 512)))


def test_oracle_stack_and_torch_path_reproduce_the_stack_built_from_the_real_reference_primitives():
    """tests/golden/pp_encoder_golden.npz was computed by tests/golden/make_golden_encoder.py with the REAL reference's sample_and_group /
    sample_and_group_all (imported /root/reference/pointnet2.py) + torch.nn Conv2d / BatchNorm2d / ReLU / max: the oracle stack
    (oracle/setabstraction_ref.py) and the module's differentiable torch path must reproduce its samples exactly and its features to
    float32 rounding."""
    from catgrasp_amd import pointnet2 as p2
    g, sd, x, start = _encoder_golden()
    xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
    nx1, r1, f1, _ = sref.sa_forward(xyz, feats, 96, 0.25, 16, sref.layers_of(sd, 'sa1.', 3), start[0])
    nx2, r2, f2, _ = sref.sa_forward(nx1, r1, 24, 0.5, 32, sref.layers_of(sd, 'sa2.', 3), start[1])
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 3))
    assert torch.equal(f1, torch.from_numpy(g['fps1'])) and torch.equal(f2, torch.from_numpy(g['fps2']))
    assert torch.equal(nx1, torch.from_numpy(g['l1_xyz'])) and torch.equal(nx2, torch.from_numpy(g['l2_xyz']))
    for got, key in ((r1, 'l1_points'), (r2, 'l2_points'), (r3, 'global_feat')):
        assert (got - torch.from_numpy(g[key])).abs().max().item() <= 2e-6, key
    enc = p2.PointNet2Encoder(**ENC_GOLDEN_CFG)
    enc.load_state_dict(sd); enc.eval()
    with torch.enable_grad():
        gf, ((x1, p1), (x2, p2_)) = enc(x, start=start)
    assert torch.equal(x1, torch.from_numpy(g['l1_xyz'])) and torch.equal(x2, torch.from_numpy(g['l2_xyz']))
    assert (p1.detach() - torch.from_numpy(g['l1_points'])).abs().max().item() <= 2e-6
    assert (gf.detach() - torch.from_numpy(g['global_feat'])).abs().max().item() <= 2e-6


def test_oracle_stack_reproduces_the_full_size_20k_point_golden_of_the_real_reference_primitives():
    """tests/golden/pp_encoder_20k_golden.npz (the real reference's sample_and_group at N = 20,000, S = 512 / 128, then
    sample_and_group_all): the oracle stack reproduces its samples and neighbour lists exactly and its features to float32 rounding,
    so the 20k-point GPU tests that use the oracle stand on the reference itself."""
    g, enc, x, start = _encoder_golden_20k()
    sd = enc.state_dict()
    xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
    nx1, r1, f1, i1 = sref.sa_forward(xyz, feats, 512, 0.2, 32, sref.layers_of(sd, 'sa1.', 3), start[0])
    nx2, r2, f2, i2 = sref.sa_forward(nx1, r1, 128, 0.4, 64, sref.layers_of(sd, 'sa2.', 3), start[1])
    r3 = sref.sa_all_forward(nx2, r2, sref.layers_of(sd, 'sa3.', 3))
    assert torch.equal(f1, torch.from_numpy(g['fps1']).long()) and torch.equal(f2, torch.from_numpy(g['fps2']).long())
    assert torch.equal(i1, torch.from_numpy(g['idx1']).long()) and torch.equal(i2, torch.from_numpy(g['idx2']).long())
    for got, key in ((r1, 'l1_points'), (r2, 'l2_points'), (r3, 'global_feat')):
        assert (got - torch.from_numpy(g[key])).abs().max().item() <= 2e-6, key


def test_prepare_start_and_kernel_routing_host_logic():
    """Host logic of the round-5 additions that needs no GPU: the FPS start preparation (the reference's torch.randint draw under a seed,
    validation) and the routing of set-abstraction shapes to the kernel families."""
    import numpy as np
    import pytest
    from catgrasp_amd import primitives as prim
    torch.manual_seed(42)
    want = torch.randint(0, 700, (3,), dtype=torch.long)          # pointnet2.py:66
    torch.manual_seed(42)
    got = prim.prepare_start(None, 3, 700, 'cpu')
    assert torch.equal(got, want) and got.dtype == torch.int64
    assert torch.equal(prim.prepare_start([1, 2, 699], 3, 700, 'cpu'), torch.tensor([1, 2, 699]))
    for bad in ([1, 2], [0, 1, 700], [-1, 0, 0]):
        with pytest.raises(ValueError):
            prim.prepare_start(bad, 3, 700, 'cpu')
    rng = np.random.default_rng(1)

    def W(cin, widths, **kw):
        prev, layers = cin, []
        for c in widths:
            layers.append((rng.normal(size=(c, prev)), rng.normal(size=c), None)); prev = c
        return prim.SetAbstractionWeights(layers, cin, 'cpu', **kw)
    assert W(9, [64, 64, 128]).kind == 'reg'                       # the register-resident kernel's shapes
    assert W(6, [32, 32, 64, 64]).kind == 'reg'                    # narrow odd nets: its LDS-strip fallback
    assert W(9, [64, 128, 256]).kind == 'tile' and W(9, [64, 96, 128]).kind == 'tile'
    assert W(131, [128, 128, 256]).kind == 'tile' and W(643, [256, 512, 1024]).cin[0] == 648
    assert W(9, [64, 128, 256], kind='reg').kind == 'reg'          # still reachable on request
    with pytest.raises(ValueError):
        W(131, [128, 128, 256], kind='reg')
    with pytest.raises(NotImplementedError):
        W(9, [64, 100])
