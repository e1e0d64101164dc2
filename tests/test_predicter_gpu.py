"""GPU parity of the predicter surface: GraspPredicter.predict_batch / NunocsPredicter.predict_nocs and the
pointnet2 nn.Module drop-ins vs the oracle (transform restatement + fp32 network restatement)."""
import numpy as np
import pytest
import torch

from catgrasp_amd import synth
from oracle import pointnet_ref as oref
from oracle import transforms_ref as tref

pytestmark = pytest.mark.gpu


def _obj(seed=0, n=2500):
    objs = synth.make_scene(1, n, seed)
    return objs[0]


@pytest.mark.parametrize('with_norm,n_cloud', [(False, 2500), (True, 2500), (True, 1200)])
def test_predict_batch_matches_reference_semantics(cuda_device, with_norm, n_cloud):
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
    ob = _obj(1, n_cloud)
    sd = synth.make_state_dict('cls', 6, 10, seed=3)
    rng = np.random.default_rng(2)
    norm = {'mean': rng.normal(0, 0.002, 6), 'std': rng.uniform(0.004, 0.3, 6)} if with_norm else None
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, normalizer=norm, device=cuda_device)
    assert gp.cfg['n_pts'] == 2048 and hasattr(gp, 'dataset') and hasattr(gp, 'model')
    P = synth.make_candidates(ob, 37, rng)
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    # reference RNG semantics: one np.random.choice per pose from the global generator
    np.random.seed(42)
    ret = gp.predict_batch(data, list(P))
    np.random.seed(42)
    xs = []
    for i in range(len(P)):
        ids = tref.draw_ids(len(ob['xyz']), 2048)
        xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids,
                                       None if norm is None else norm['mean'], None if norm is None else norm['std'])['input'])
    x = torch.from_numpy(np.stack(xs)).float()            # predicter.py:84 casts the float64 input to float32
    logits, _ = oref.pointnet_cls_forward(sd, x)
    ref = tref.predict_batch_post(logits.numpy())
    assert len(ret) == len(ref) == 37
    for a, b in zip(ret, ref):
        assert np.abs(a[2] - b[2]).max() <= 1e-4
        assert abs(float(a[1]) - float(b[1])) <= 1e-4
        top2 = np.sort(b[2])[-2:]
        if top2[1] - top2[0] > 2e-4:
            assert int(a[0]) == int(b[0])
    pg = tref.p_G(np.stack([r[2] for r in ret]), 10)
    assert np.abs(pg - tref.p_G(np.stack([r[2] for r in ref]), 10)).max() <= 1e-4
    assert gp.predict_batch(data, []) == []


def test_build_grasp_input_kernel(cuda_device):
    """The device GraspDataset.transform alone, against the float64 restatement (dataset_grasp.py:63-91)."""
    from catgrasp_amd import ops, transforms
    ob = _obj(5)
    rng = np.random.default_rng(6)
    P = synth.make_candidates(ob, 9, rng)
    ids = np.stack([rng.choice(len(ob['xyz']), 2048, replace=False) for _ in P]).astype(np.int32)
    mean = rng.normal(0, 0.002, 6); std = rng.uniform(0.004, 0.3, 6)
    dc = transforms.DeviceCloud(ob['xyz'], ob['normal'], cuda_device)
    pinv = torch.from_numpy(transforms.pose_inverse_rows(P, dc.center)).to(cuda_device)
    m, s = transforms.normalizer_device({'mean': mean, 'std': std}, cuda_device)
    x = ops.build_grasp_input(dc.xyz, dc.normal, torch.from_numpy(ids).to(cuda_device), pinv, m, s).cpu().numpy()
    ref = np.stack([tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids[i], mean, std)['input'] for i in range(len(P))])
    # float32 evaluation of a float64 formula: ~1e-6 of the value range
    assert np.abs(x - ref).max() <= 2e-5 * np.abs(ref).max()


def test_nunocs_predict_nocs(cuda_device):
    from catgrasp_amd.predicter import DEFAULT_NUNOCS_CFG, NunocsPredicter
    ob = _obj(8, 3000)
    sd = synth.make_state_dict('seg', 6, 300, seed=4)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=sd, device=cuda_device)
    assert npred.min_scale == [0.005, 0.005, 0.001]
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    np.random.seed(7)
    nocs, conf, dt = npred.predict_nocs(data)
    np.random.seed(7)
    ids = tref.draw_ids(len(ob['xyz']), 8192)
    tr = tref.nunocs_transform(ob['xyz'].copy(), ob['normal'].copy(), ids)
    assert np.array_equal(dt['keep_ids'], tr['keep_ids']) and np.array_equal(dt['cloud_xyz_original'], tr['cloud_xyz_original'])
    logits, _ = oref.pointnet_seg_forward(sd, torch.from_numpy(tr['input'][None]).float())
    rcoords, rconf = tref.nunocs_decode(logits[0].numpy(), 100)
    assert nocs.shape == (8192, 3) and nocs.dtype == np.float32
    # coordinates equal wherever the top-2 bin logits are separated by more than the tolerance
    lg = logits[0].numpy().reshape(-1, 3, 100)
    srt = np.sort(lg, axis=-1)
    clear = (srt[..., -1] - srt[..., -2]) > 2e-4
    assert clear.mean() > 0.99
    assert np.array_equal(nocs[clear], rcoords[clear])
    zclear = clear[:, 2]
    assert np.abs(conf[zclear] - rconf[zclear]).max() <= 1e-4
    # predict(): the device RANSAC replayed through the oracle on the same numpy stream (tests/test_aligning_gpu.py holds the helpers)
    from test_aligning_gpu import _oracle_predict_tail, _same_outcome
    np.random.seed(9)
    out = npred.predict(data)
    np.random.seed(9)
    nocs2, _, dt2 = npred.predict_nocs(data)
    _same_outcome(out, _oracle_predict_tail(nocs2, dt2['cloud_xyz_original'], npred.min_scale, npred.max_scale, 10000), npred)


def test_pointnet2_modules_dropin(cuda_device):
    """pointnet2.PointNetCls / PointNetSeg as nn.Modules: reference state_dict keys, eval->HIP, train->torch."""
    from catgrasp_amd import pointnet2 as p2
    sd = synth.make_state_dict('cls', 6, 10, seed=9)
    m = p2.PointNetCls(6, 10)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict({'module.' + k: v for k, v in sd.items()} if False else sd)
    x = torch.from_numpy(np.random.default_rng(1).normal(0, 0.5, (4, 300, 6)).astype(np.float32))
    ref, ref_tf = oref.pointnet_cls_forward(sd, x)
    m.cuda().eval()
    with torch.no_grad():
        y, tf = m(x.cuda())
    assert ((y.cpu() - ref).abs() / ref.abs().clamp(min=1)).max().item() <= 1e-4
    # the torch (training) path computes the same function in eval-mode statistics
    m.eval()
    with torch.enable_grad():
        y2, _ = m(x.cuda())
    assert ((y2.detach().cpu() - ref).abs() / ref.abs().clamp(min=1)).max().item() <= 1e-4
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m.cpu()(x)
    m.cuda()
    bad = x.clone(); bad[1, 7, 2] = float('nan')                # the reference would return NaN logits; here it is an error,
    with pytest.raises(ValueError):                             # never a silently max-pooled-away point
        with torch.no_grad():
            m(bad.cuda())
    with pytest.raises(ValueError):
        GraspPredicterData = {'cloud_xyz': np.full((100, 3), np.inf), 'cloud_normal': np.zeros((100, 3))}
        from catgrasp_amd import transforms
        transforms.DeviceCloud(GraspPredicterData['cloud_xyz'], GraspPredicterData['cloud_normal'], cuda_device)
    ss = synth.make_state_dict('seg', 6, 300, seed=10)
    s = p2.PointNetSeg(6, 300)
    s.load_state_dict(ss)
    s.cuda().eval()
    with torch.no_grad():
        ys, _ = s(x.cuda())
    rs, _ = oref.pointnet_seg_forward(ss, x)
    assert ((ys.cpu() - rs).abs() / rs.abs().clamp(min=1)).max().item() <= 1e-4


def test_standalone_building_blocks_run_on_the_hip_passes(cuda_device, mlp_precision):
    """`from pointnet2 import *` users who call STN3d / PointNetEncoder directly in eval mode get the fused HIP passes
    (not a silent stock-torch second backend); a free-standing STNkd runs its layers on the HIP GEMM kernel."""
    from catgrasp_amd import ops
    from catgrasp_amd import pointnet2 as p2
    sd = synth.make_state_dict('seg', 6, 300, seed=12)
    x = torch.from_numpy(np.random.default_rng(2).normal(0, 0.5, (3, 6, 700)).astype(np.float32))       # (B,D,N) like the reference
    calls = []
    real = ops.pointmlp_max
    ops.pointmlp_max = lambda *a, **k: (calls.append(k.get('mid_mode', 0)), real(*a, **k))[1]
    try:
        for gf in (True, False):
            enc = p2.PointNetEncoder(global_feat=gf, feature_transform=True, channel=6)
            enc.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith('feat.')})
            enc.cuda().eval()
            calls.clear()
            with torch.no_grad():
                out, trans, tf = enc(x.cuda())
            assert calls == [0, 1, 2]                                   # the three fused passes ran, not torch conv1d
            psd = oref.prepared_state_dict({k[5:]: v for k, v in sd.items() if k.startswith('feat.')})
            ref, rtrans, rtf = oref._nn_encoder(psd, '', x, gf)
            assert out.shape == ref.shape
            for a, b in ((out, ref), (trans, rtrans), (tf, rtf)):
                assert ((a.cpu() - b).abs() / b.abs().clamp(min=1)).max().item() <= 1e-4
        stn = p2.STN3d(6)
        stn.load_state_dict({k[9:]: v for k, v in sd.items() if k.startswith('feat.stn.')})
        stn.cuda().eval()
        calls.clear()
        with torch.no_grad():
            t = stn(x.cuda())
        assert calls == [0] and t.shape == (3, 3, 3)
        rt = oref._nn_stn(oref.prepared_state_dict({k[9:]: v for k, v in sd.items() if k.startswith('feat.stn.')}), '', x, 3)
        assert (t.cpu() - rt).abs().max().item() <= 1e-4
        with torch.enable_grad():                                       # grad-enabled call: differentiable torch ops, same function
            t2 = stn(x.cuda())
        assert calls == [0] and (t2.detach().cpu() - rt).abs().max().item() <= 1e-4
    finally:
        ops.pointmlp_max = real
    kd = p2.STNkd(64).cuda().eval()              # free-standing STNkd: GEMM kernels + group max (parity: tests/test_pointnet_blocks_gpu.py)
    xk = torch.randn(2, 64, 100, device=cuda_device)
    with torch.no_grad():
        t_hip = kd(xk)
    with torch.enable_grad():
        t_torch = kd(xk)
    assert t_hip.shape == (2, 64, 64) and (t_hip - t_torch.detach()).abs().max().item() <= 1e-4


def test_dropin_predicters_against_the_real_reference_outputs(cuda_device, mlp_precision):
    """The headline parity test: catgrasp_amd.predicter run with the same numpy seed, weights and normaliser as the REAL
    reference predicter (outputs committed in tests/golden/predicter_golden.npz by make_golden_predicter.py) must return the
    same [label, confidence, probs] lists (1e-4) and the same NUNOCS cloud (away from arg-max ties)."""
    import os
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    p = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'predicter_golden.npz'))
    precision = mlp_precision
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=77),
                        normalizer={'mean': p['mean'], 'std': p['std']}, device=cuda_device)
    np.random.seed(123)
    ret = gp.predict_batch({'cloud_xyz': p['xyz'], 'cloud_normal': p['normal']}, list(p['poses']))
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=78), device=cuda_device)
    np.random.seed(321)
    nocs, conf, dt = npred.predict_nocs({'cloud_xyz': p['xyz'], 'cloud_normal': p['normal']})
    assert len(ret) == len(p['poses'])
    assert np.abs(np.array([r[2] for r in ret]) - p['grasp_probs']).max() <= 1e-4
    assert np.abs(np.array([float(r[1]) for r in ret]) - p['grasp_conf']).max() <= 1e-4
    gap = np.sort(p['grasp_probs'], axis=1)
    sure = (gap[:, -1] - gap[:, -2]) > 2e-4
    assert np.array_equal(np.array([int(r[0]) for r in ret])[sure], p['grasp_labels'][sure])
    assert np.array_equal(dt['keep_ids'], p['nocs_keep_ids'])
    clear = p['nocs_top2_gap'] > (1e-3 if precision in ('bf16x3', 'f16fp8x2') else 2e-4)
    assert clear.mean() > 0.98 and np.array_equal(nocs[clear], p['nocs_cloud'][clear])
    zc = clear[:, 2]
    assert np.abs(conf[zc] - p['nocs_conf_z'][zc]).max() <= 1e-4


def _rel_logit_err(gp, sd, ob, P, ids):
    """max |logits - f64 oracle| / max |oracle| of the product's nn.Module forward on one candidate batch"""
    x = np.stack([tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids[i])['input'] for i in range(len(P))])
    xt = torch.from_numpy(x).float()
    ref = oref.pointnet_cls_forward(sd, xt, torch.float64)[0]
    with torch.no_grad():
        y = gp.model(xt.to(gp.device))[0].cpu().double()
    assert torch.isfinite(y).all()
    return float((y - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('case', ['activations_1e6', 'weights_beyond_half', 'activations_subnormal', 'weights_tiny'])
def test_half_range_guard_falls_back_instead_of_failing(cuda_device, mlp_precision, case):
    """the split-half arithmetic ('f16x3', the default) has IEEE-half's exponent range.  A checkpoint whose
    activations or folded weights leave it -- upwards (>= 65504) or downwards (whole layer outputs / weight matrices below 2^-6,
    where the low pieces sink into the half subnormals) -- must neither raise nor return damaged scores: the kernels report the
    condition in a per-call status word and the engine re-evaluates with bf16 pieces (float32's exponent range); layers whose
    weight images fail the pre-screen never use the half kernels.  All four nets compute (up to rounding) ordinary functions;
    every arithmetic mode must stay within 1e-4 (relative to the logit scale) of the float64 evaluation."""
    import warnings
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
    sd = {k: v.clone() for k, v in synth.make_state_dict('cls', 6, 10, seed=5).items()}
    if case == 'activations_1e6':
        sd['feat.conv1.weight'] *= 1e6
    elif case == 'weights_beyond_half':
        sd['feat.bn3.weight'] *= 3e6            # folded 128->1024 weights ~ 3e6 * 0.14: not representable in half
        sd['fc1.weight'] /= 3e6                 # ... and the global feature it produces (~1e6) overflows the next layer's split
    elif case == 'activations_subnormal':
        # ReLU is positively homogeneous: scaling one layer's output by 1e-5 and the consumers' weights by 1e5 leaves the function
        # unchanged, but every activation of that layer is now < 2^-6 (most of them half subnormals)
        sd['feat.bn1.weight'] *= 1e-5; sd['feat.bn1.bias'] *= 1e-5
        sd['feat.fstn.conv1.weight'] *= 1e5; sd['feat.conv2.weight'] *= 1e5
    else:
        sd['feat.stn.bn2.weight'] *= 1e-4; sd['feat.stn.bn2.bias'] *= 1e-4       # folded 64->128 weights ~ 1e-5: all below 2^-6
        sd['feat.stn.conv3.weight'] *= 1e4                                       # same function (ReLU homogeneity)
    ob = synth.make_scene(1, 1500, seed=4)[0]
    P = synth.make_candidates(ob, 8, np.random.default_rng(1))
    ids = np.stack([np.random.default_rng(i).choice(len(ob['xyz']), 2048, replace=True) for i in range(len(P))])
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=cuda_device)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        assert _rel_logit_err(gp, sd, ob, P, ids) <= 1e-4
        ret = gp.predict_batch({'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}, list(P), ids=ids)
    assert len(ret) == 8 and all(np.isfinite(r[2]).all() and abs(float(r[2].sum()) - 1) < 1e-5 for r in ret)
    if mlp_precision in ('f16x3', 'f16fp8x2') and case in ('weights_beyond_half', 'weights_tiny'):
        assert not all(gp._W.half_ok.values())          # the pre-screen took those layers off the half kernels


def test_predict_batch_device_rng_and_id_validation(cuda_device):
    """rng='device' draws the per-pose resampling on the device (no host loop); it is reproducible under np.random.seed, every row
    is a duplicate-free subset when the cloud has >= n_pts points, and the scores follow the same oracle.  Explicit ids are range
    checked like the reference's numpy indexing."""
    from catgrasp_amd import transforms
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
    sd = synth.make_state_dict('cls', 6, 10, seed=5)
    ob = synth.make_scene(1, 2500, seed=4)[0]
    P = synth.make_candidates(ob, 40, np.random.default_rng(1))
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=cuda_device)
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    np.random.seed(7); a = gp.predict_batch(data, list(P), rng='device')
    np.random.seed(7); b = gp.predict_batch(data, list(P), rng='device')
    assert all(np.array_equal(x[2], y[2]) for x, y in zip(a, b))
    ids = transforms.draw_ids_device(2500, 2048, 300, cuda_device, seed=11).cpu().numpy()
    assert ids.min() >= 0 and ids.max() < 2500 and all(len(np.unique(r)) == 2048 for r in ids)
    assert len({r.tobytes() for r in ids}) == 300
    # every index is (close to) equally likely and every output slot is uniform: chi-square-free sanity bounds
    cnt = np.bincount(ids.reshape(-1), minlength=2500)
    assert 200 < cnt.min() and cnt.max() <= 300          # binomial(300, 0.82) per index: mean 245.8, sd 6.7
    assert abs(ids[:, 0].mean() - 1249.5) < 150 and abs(ids[:, -1].mean() - 1249.5) < 150
    # the rows ARE the keyed permutation of oracle/draw_bijection_ref.py, bit for bit: odd sizes, a power of four (no cycle walking), a
    # cloud beyond the old 65,535-point limit, a 64-bit seed, a base offset, and a shard (row_offset) == the rows of the whole
    from oracle import draw_bijection_ref as dref
    assert np.array_equal(ids, dref.draw_rows(2500, 2048, 300, 11))
    for nv, npts, cnt, seed, base, off in ((4096, 2048, 9, 3, 0, 0), (2049, 2048, 5, 2 ** 40 + 77, 10000, 0), (70000, 2048, 4, 5, 0, 2 ** 33),
                                            (1025, 1000, 7, 1, 3, 0), (20000, 8192, 3, 9, 0, 41)):
        got = transforms.draw_ids_device(nv, npts, cnt, cuda_device, seed=seed, base=base, row_offset=off).cpu().numpy()
        assert np.array_equal(got, dref.draw_rows(nv, npts, cnt, seed, base=base, row_offset=off)), (nv, npts)
        assert all(len(np.unique(r)) == npts for r in got)
    whole = transforms.draw_ids_device(2500, 2048, 40, cuda_device, seed=21).cpu().numpy()
    part = transforms.draw_ids_device(2500, 2048, 15, cuda_device, seed=21, row_offset=25).cpu().numpy()
    assert np.array_equal(whole[25:], part)
    small = transforms.draw_ids_device(700, 512, 20, cuda_device, seed=2).cpu().numpy()        # <= 1,024 points: the sort kernel
    assert all(len(np.unique(r)) == 512 for r in small) and small.max() < 700
    rep = transforms.draw_ids_device(700, 2048, 50, cuda_device, seed=3).cpu().numpy()       # with replacement
    assert rep.min() >= 0 and rep.max() < 700 and rep.shape == (50, 2048)
    ref = tref.predict_batch_post(oref.pointnet_cls_forward(sd, torch.from_numpy(np.stack(
        [tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids[i])['input'] for i in range(len(P))])).float())[0].numpy())
    got = gp.predict_batch(data, list(P), ids=ids[:len(P)])
    assert max(float(np.abs(g[2] - r[2]).max()) for g, r in zip(got, ref)) <= 1e-4
    bad = ids[:len(P)].copy(); bad[3, 5] = 2500
    with pytest.raises(IndexError):
        gp.predict_batch(data, list(P), ids=bad)
    # poses np.linalg.inv (dataset_grasp.py:69-70) would refuse or that the device's closed-form affine inverse cannot represent:
    # a singular pose raises LinAlgError like the reference, a non-affine last row and a NaN are refused -- none is pooled away into
    # finite-looking probabilities; numpy's generator is left untouched by the failed call (all poses are converted before the draw)
    for bad_pose, exc in ((np.diag([1.0, 1.0, 0.0, 1.0]), np.linalg.LinAlgError), (np.zeros((4, 4)), (np.linalg.LinAlgError, ValueError)),
                          (np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0.5, 1.0]]), ValueError),
                          (np.full((4, 4), np.nan), ValueError)):
        for rng_mode in ('device', 'numpy'):
            Pb = [p.copy() for p in P]; Pb[17] = bad_pose
            with pytest.raises(exc):
                gp.predict_batch(data, Pb, rng=rng_mode)
    np.random.seed(3); st = np.random.get_state()[1].copy()
    with pytest.raises(ValueError):
        gp.predict_batch(data, list(P[:30]) + [np.eye(3)], rng='numpy')         # malformed pose late in the list
    assert np.array_equal(np.random.get_state()[1], st)
