"""CPU tests (no GPU): pin the oracle restatement (oracle/pointnet_ref.py) to the REAL reference.
 * against tests/golden/pointnet2_golden.npz, produced by tests/golden/make_golden.py from the imported
   /root/reference/pointnet2.py (always runs);
 * live against /root/reference when it is present (build container only)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from catgrasp_amd import synth
from oracle import pointnet_ref as oref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # tests/fps_clouds.py
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pointnet2_golden.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize('tag', ['cls_101', 'cls_102', 'seg_103'])
def test_models_match_reference_golden(gold, tag):
    kind = tag.split('_')[0]
    seed, gain, n_out = gold[tag + '_meta']
    sd = synth.make_state_dict(kind, 6, int(n_out), seed=int(seed), gain=float(gain))
    x = torch.from_numpy(gold[tag + '_x'])
    if kind == 'cls':
        y, tf = oref.pointnet_cls_forward(sd, x)
        y = y.numpy()
    else:
        y, tf = oref.pointnet_seg_forward(sd, x)
        y = y.numpy()[:, ::3, ::4]
    ref = gold[tag + '_y']
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(tf.numpy()[:, ::7, ::5] - gold[tag + '_tf']).max() <= 2e-5
    # float64 evaluation agrees too (bounds the float32 rounding of both)
    y64 = (oref.pointnet_cls_forward if kind == 'cls' else oref.pointnet_seg_forward)(sd, x, torch.float64)[0].numpy()
    y64 = y64 if kind == 'cls' else y64[:, ::3, ::4]
    assert np.abs(y64 - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    # the torch-nn-ops formulation (what bench.py's cpu_baseline times) is pinned to the same golden outputs
    psd = oref.prepared_state_dict(sd)
    with torch.no_grad():
        yn, tfn = (oref.pointnet_cls_forward_nnops if kind == 'cls' else oref.pointnet_seg_forward_nnops)(psd, x)
    yn = yn.numpy() if kind == 'cls' else yn.numpy()[:, ::3, ::4]
    assert np.abs(yn - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(tfn.numpy()[:, ::7, ::5] - gold[tag + '_tf']).max() <= 2e-5


def test_primitives_match_reference_golden(gold):
    xyz = torch.from_numpy(gold['prim_xyz']); feats = torch.from_numpy(gold['prim_feats'])
    assert np.abs(oref.square_distance(xyz[:, :16], xyz).numpy() - gold['sqdist']).max() <= 1e-6
    assert np.array_equal(oref.index_points(feats, torch.from_numpy(gold['index_idx'])).numpy(), gold['index_out'])
    torch.manual_seed(77)
    start = torch.randint(0, 700, (2,), dtype=torch.long)          # pointnet2.py:66
    fps = oref.farthest_point_sample(xyz, 48, start)
    assert np.array_equal(fps.numpy(), gold['fps'])
    new_xyz = oref.index_points(xyz, fps)
    assert np.array_equal(oref.query_ball_point(0.03, 16, xyz, new_xyz).numpy(), gold['ball'])
    torch.manual_seed(78)
    start = torch.randint(0, 700, (2,), dtype=torch.long)
    nx, npnts, gx, fi = oref.sample_and_group(32, 0.04, 8, xyz, feats, start)
    assert np.array_equal(fi.numpy(), gold['sg_fps']) and np.array_equal(nx.numpy(), gold['sg_new_xyz'])
    assert np.array_equal(gx.numpy(), gold['sg_grouped_xyz']) and np.array_equal(npnts.numpy(), gold['sg_new_points'])
    ax, ap = oref.sample_and_group_all(xyz, feats)
    assert np.array_equal(ax.numpy(), gold['sga_new_xyz'])
    assert abs(float(ap.numpy().astype(np.float64).sum()) - float(gold['sga_new_points_sum'][0])) < 1e-9


def test_farthest_point_sample_matches_reference_golden_at_working_sizes():
    """tests/golden/fps_large_golden.npz: the real reference's samples on 3,000 .. 24,000-point clouds (filled volume, surface,
    duplicated points, lattice) -- the sizes and the tie situations the blob-skipping HIP kernel is built for."""
    import fps_clouds
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fps_large_golden.npz'))
    for kind, n, seed in fps_clouds.CASES:
        xyz = torch.from_numpy(fps_clouds.cloud(kind, n, seed)[None])
        torch.manual_seed(seed)
        start = torch.randint(0, n, (1,), dtype=torch.long)        # pointnet2.py:66
        assert np.array_equal(start.numpy(), gold[f'{kind}_{n}_start'])
        got = oref.farthest_point_sample(xyz, fps_clouds.NPOINT, start).numpy()
        assert np.array_equal(got, gold[f'{kind}_{n}_fps']), (kind, n)


@pytest.mark.skipif(not os.path.exists('/root/reference/pointnet2.py'), reason='reference checkout only exists in the build container')
def test_live_against_imported_reference():
    for m in ('cv2', 'torchvision'):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, '/root/reference')
    try:
        import pointnet2 as ref
    finally:
        sys.path.remove('/root/reference')
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.normal(0, 0.5, (2, 200, 6)).astype(np.float32))
    for kind, n_out, cls in (('cls', 10, ref.PointNetCls), ('seg', 300, ref.PointNetSeg)):
        sd = synth.make_state_dict(kind, 6, n_out, seed=31)
        m = cls(6, n_out); m.load_state_dict(sd); m.eval()
        with torch.no_grad():
            y, tf = m(x)
        y2, tf2 = (oref.pointnet_cls_forward if kind == 'cls' else oref.pointnet_seg_forward)(sd, x)
        assert (y - y2).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
        assert (tf - tf2).abs().max().item() <= 2e-5
    # our drop-in nn.Modules expose exactly the reference's parameter / buffer names and shapes
    sys.modules.pop('pointnet2', None)
    from catgrasp_amd import pointnet2 as ours
    for (a, b) in ((ours.PointNetCls(6, 10), ref.PointNetCls(6, 10)), (ours.PointNetSeg(6, 300), ref.PointNetSeg(6, 300))):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)


@pytest.mark.skipif(not os.path.exists('/root/reference/pointnet2.py'), reason='reference checkout only exists in the build container')
def test_training_mode_equals_the_imported_reference_bit_for_bit():
    """trainer_grasp.py / trainer_nunocs.py must keep working on the drop-in modules.  With the same state_dict, the
    same input and the same torch seed (Dropout p=0.4 in PointNetCls), the drop-in's train-mode forward, the gradients of every
    parameter and the BatchNorm running statistics after the step are IDENTICAL to the reference modules'."""
    for m in ('cv2', 'torchvision'):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, '/root/reference')
    try:
        sys.modules.pop('pointnet2', None)
        import pointnet2 as ref
    finally:
        sys.path.remove('/root/reference')
        sys.modules.pop('pointnet2', None)
    from catgrasp_amd import pointnet2 as ours
    x = torch.from_numpy(np.random.default_rng(8).normal(0, 0.5, (4, 120, 6)).astype(np.float32))
    for kind, n_out, name in (('cls', 10, 'PointNetCls'), ('seg', 30, 'PointNetSeg')):
        sd = synth.make_state_dict(kind, 6, n_out, seed=3)
        a, b = getattr(ours, name)(6, n_out), getattr(ref, name)(6, n_out)
        a.load_state_dict(sd); b.load_state_dict(sd)
        a.train(); b.train()
        torch.manual_seed(0); ya, ta = a(x.clone())
        torch.manual_seed(0); yb, tb = b(x.clone())
        assert torch.equal(ya, yb) and torch.equal(ta, tb)
        ya.square().sum().backward(); yb.square().sum().backward()
        gb = dict(b.named_parameters())
        assert all(torch.equal(p.grad, gb[k].grad) for k, p in a.named_parameters())
        sb = b.state_dict()
        assert all(torch.equal(v, sb[k]) for k, v in a.state_dict().items())


def test_dropin_building_blocks_with_every_constructor_setting_match_the_reference():
    """The drop-in modules in their torch formulation (grad enabled: the path the trainers use; CPU) vs the REAL reference classes for
    the constructor arguments the live pipeline does not use -- STN3d(3|5), STNkd(64|20), PointNetEncoder(global_feat x
    feature_transform x channel 3|4|6) incl. the reference defaults, PointNetCls(3,10), PointNetSeg(4,30)
    (tests/golden/pointnet2_blocks_golden.npz).  Same state_dict keys, same numbers.  The HIP path of the same modules is checked
    against the same file on the GPU (tests/test_pointnet_blocks_gpu.py)."""
    from catgrasp_amd import pointnet2 as p2
    g = np.load(os.path.join(os.path.dirname(GOLD), 'pointnet2_blocks_golden.npz'))

    def sample(t):
        if t.dim() == 3 and t.shape[1] == 1088:
            return t[:, ::9, ::4]
        if t.dim() == 3 and tuple(t.shape[1:]) == (64, 64):
            return t[:, ::3, ::3]
        return t
    cases = [(f'stn3d_c{c}', lambda c=c: p2.STN3d(c)) for c in (3, 5)] + [(f'stnkd_k{k}', lambda k=k: p2.STNkd(k=k)) for k in (64, 20)]
    cases += [(f'enc_g{int(gf)}_f{int(ft)}_c{c}', lambda gf=gf, ft=ft, c=c: p2.PointNetEncoder(global_feat=gf, feature_transform=ft, channel=c))
              for gf in (True, False) for ft in (False, True) for c in (3, 4, 6)]
    cases += [('cls_c3', lambda: p2.PointNetCls(3, 10)), ('seg_c4', lambda: p2.PointNetSeg(4, 30))]
    for tag, make in cases:
        m = make()
        m.load_state_dict(synth.seeded_like(m.state_dict(), int(g[tag + '_seed'][0])))
        m.eval()
        with torch.enable_grad():
            y = m(torch.from_numpy(g[tag + '_x']))
        ys = y if isinstance(y, tuple) else (y,)
        for i, t in enumerate(ys):
            ref = g[f'{tag}_y{i}']
            if t is None:
                assert ref.size == 0, tag
                continue
            got = sample(t.detach()).numpy()
            assert got.shape == ref.shape, (tag, i)
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (tag, i)
