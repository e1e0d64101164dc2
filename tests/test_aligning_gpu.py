"""GPU parity of the 9-D similarity RANSAC (aligning.estimate9DTransform) vs the numpy restatement."""
import numpy as np
import pytest

from catgrasp_amd import synth
from oracle import aligning_ref

pytestmark = pytest.mark.gpu


def _problem(seed, n=3000, outliers=0.2, noise=0.0001):
    rng = np.random.default_rng(seed)
    nocs = rng.uniform(-0.5, 0.5, (n, 3))
    R = synth.random_rotation(rng); s = np.array([0.016, 0.02, 0.007]); t = np.array([0.02, -0.03, 0.62])
    T = np.eye(4); T[:3, :3] = R @ np.diag(s); T[:3, 3] = t
    obs = nocs @ T[:3, :3].T + t + rng.normal(0, noise, (n, 3))
    bad = rng.random(n) < outliers
    nocs_pred = nocs.copy()
    nocs_pred[bad] = rng.uniform(-0.5, 0.5, (bad.sum(), 3))       # wrong NUNOCS predictions
    return nocs_pred, obs, T


def test_hypotheses_and_best_transform_match_oracle(cuda_device):
    import ctypes
    import torch
    from catgrasp_amd import aligning
    from catgrasp_amd import _lib as L
    src, dst, T_true = _problem(0)
    rng = np.random.default_rng(1)
    ids = np.stack([rng.choice(len(src), 4, replace=False) for _ in range(600)]).astype(np.int32)
    ids[5] = [7, 7, 8, 9]                                         # repeated sample -> degenerate, rejected
    min_s, max_s, max_d = [0.005, 0.005, 0.001], [0.05, 0.05, 0.05], np.array([1.2, 1.2, 1.2])
    T_ref, inl_ref, outs = aligning_ref.estimate9DTransform(src, dst, 0.003, ids, max_s, min_s, max_d)
    # per-hypothesis agreement (accept/reject and inlier counts)
    dev = cuda_device
    d_src = torch.from_numpy(src).to(dev); d_dst = torch.from_numpy(dst).to(dev); d_ids = torch.from_numpy(ids).to(dev)
    counts = torch.empty((len(ids),), dtype=torch.int32, device=dev)
    transforms = torch.empty((len(ids), 16), dtype=torch.float64, device=dev)
    D3 = ctypes.c_double * 3
    L.check(L.lib().cg_ransac_9d(L._p(d_src), L._p(d_dst), ctypes.c_int(len(src)), L._p(d_ids), ctypes.c_int(len(ids)), ctypes.c_double(0.003),
                                 D3(*min_s), D3(*max_s), D3(*max_d), L._p(counts), L._p(transforms), L._stream()), 'ransac')
    c = counts.cpu().numpy(); Ts = transforms.cpu().numpy().reshape(-1, 4, 4)
    ref_c = np.array([-1 if o is None else o[0] for o in outs])
    assert (ref_c >= 0).sum() > 20 and (ref_c < 0).sum() > 20 and ref_c[5] == -1
    assert np.array_equal(c, ref_c), f'{(c != ref_c).sum()} hypotheses disagree'
    for i in np.flatnonzero(ref_c >= 0)[:50]:
        assert np.abs(Ts[i] - outs[i][1]).max() < 1e-9
    T, inl = aligning.estimate9DTransform(src, dst, 0.003, max_iter=len(ids), max_scale=max_s, min_scale=min_s, max_dimensions=max_d, ids=ids)
    assert np.abs(T - T_ref).max() < 1e-9 and np.array_equal(inl, inl_ref)
    # and it actually recovers the pose
    assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < 2e-3
    assert np.abs(np.linalg.norm(T[:3, :3], axis=0) - np.array([0.016, 0.02, 0.007])).max() < 2e-3


def test_reference_rng_semantics_and_failure_modes(cuda_device):
    from catgrasp_amd import aligning
    src, dst, _ = _problem(3, n=800)
    np.random.seed(11)
    T1, i1 = aligning.estimate9DTransform(src, dst, 0.003, max_iter=300, max_scale=[0.05] * 3, min_scale=[0.001] * 3)
    np.random.seed(11)
    ids = np.stack([np.random.choice(len(src), size=4, replace=False) for _ in range(300)])   # aligning.py:89-93
    T2, i2, _ = aligning_ref.estimate9DTransform(src, dst, 0.003, ids, [0.05] * 3, [0.001] * 3)
    assert np.abs(T1 - T2).max() < 1e-9 and np.array_equal(i1, i2)
    # impossible scale bounds -> (None, None) like aligning.py:103-104
    assert aligning.estimate9DTransform(src, dst, 0.003, max_iter=50, max_scale=[1e-6] * 3, min_scale=[0] * 3) == (None, None)
    with pytest.raises(ValueError):              # the kd-tree evaluation needs its voxel size (aligning.py:65,69)
        aligning.estimate9DTransform(src, dst, 0.003, max_iter=5, use_kdtree_for_eval=True)


def test_kdtree_evaluation_branch(cuda_device):
    """use_kdtree_for_eval=True (aligning.py:63-76): two-sided nearest-neighbour errors against voxel-down-sampled clouds.  Against the
    golden of the REAL worker (tests/golden/aligning_kd_golden.npz; cv2 / open3d substituted as stated there): per-hypothesis ratios
    through the product's own pieces, and the call's (transform, inliers) == the arg-max over the golden's ratios."""
    import os
    import torch
    from catgrasp_amd import aligning
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'aligning_kd_golden.npz'))
    src, dst, ids = g['src'], g['dst'], g['ids'].astype(np.int32)
    min_s, max_s, max_d = [0.005, 0.005, 0.001], [0.05, 0.05, 0.05], np.array([1.2, 1.2, 1.2])
    d_src, d_dst = torch.from_numpy(src).to(cuda_device), torch.from_numpy(dst).to(cuda_device)
    for si, (thr, res) in enumerate(g['settings']):
        ratio, tfs, inl = g[f'ratio{si}'], g[f'tf{si}'], g[f'inliers{si}']
        acc = np.flatnonzero(ratio >= 0)
        # the down-sampled target == the oracle's (as a set of centroids)
        a = aligning_ref.voxel_down_sample(dst, res)
        b = aligning.voxel_down_sample_device(d_dst, float(res)).cpu().numpy()
        assert a.shape == b.shape and np.abs(a[np.lexsort(a.T)] - b[np.lexsort(b.T)]).max() < 1e-12
        # every accepted hypothesis scored by the product's evaluation alone reproduces the golden ratio and inlier mask
        for h in acc:
            T, inliers = aligning._kdtree_eval(d_src, d_dst, torch.from_numpy(tfs[h].reshape(1, 16)).to(cuda_device), [0], float(thr), float(res))
            m = np.zeros(len(src), dtype=np.uint8); m[inliers] = 1
            assert np.array_equal(m, inl[h]) and np.abs(T - tfs[h]).max() == 0
            d1 = aligning._nn_dist((d_src @ torch.from_numpy(tfs[h][:3, :3].T.copy()).to(cuda_device) + torch.from_numpy(tfs[h][:3, 3].copy()).to(cuda_device)).contiguous(),
                                   aligning.voxel_down_sample_device(d_dst, float(res)).contiguous())
            assert int((d1 <= thr).sum()) == int(inl[h].sum())
        T, inliers = aligning.estimate9DTransform(src, dst, thr, max_iter=len(ids), use_kdtree_for_eval=True, kdtree_eval_resolution=res,
                                                  max_scale=max_s, min_scale=min_s, max_dimensions=max_d, ids=ids)
        best = acc[np.argmax(ratio[acc])]                 # aligning.py:112: first maximum over the accepted hypotheses
        m = np.zeros(len(src), dtype=np.uint8); m[inliers] = 1
        assert np.abs(T - tfs[best]).max() < 1e-9 and np.array_equal(m, inl[best])


def test_nunocs_predicter_predict_end_to_end(cuda_device):
    """NunocsPredicter.predict with the device RANSAC as align_fn: returns (nocs_cloud, 4x4) and sets the attributes
    run_grasp_simulation.py reads (best_ratio, nocs_pose)."""
    from catgrasp_amd import aligning
    from catgrasp_amd.predicter import DEFAULT_NUNOCS_CFG, NunocsPredicter
    ob = synth.make_scene(1, 3000, 2)[0]
    sd = synth.make_state_dict('seg', 6, 300, seed=4)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=sd, device=cuda_device, align_fn=aligning.estimate9DTransform)
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    np.random.seed(21)
    out = npred.predict(data)
    # The same call replayed through the oracle: same numpy stream (resample draw, then 2 x 10,000 hypothesis draws), the restated
    # RANSAC on the NUNOCS cloud the product decoded, the threshold loop / det check / ratio of predicter.py:159-203.
    np.random.seed(21)
    nocs, _, dt = npred.predict_nocs(data)
    exp = _oracle_predict_tail(nocs, dt['cloud_xyz_original'], npred.min_scale, npred.max_scale, 10000)
    _same_outcome(out, exp, npred)


def _oracle_predict_tail(nocs_cloud, ori, min_scale, max_scale, max_iter):
    best_ratio, best_transform = 0, None
    for thres in [0.003, 0.005]:
        ids = np.stack([np.random.choice(len(nocs_cloud), size=4, replace=False) for _ in range(max_iter)])      # aligning.py:89-93
        transform, _, _ = aligning_ref.estimate9DTransform(nocs_cloud.astype(np.float64), ori.astype(np.float64), thres, ids, max_scale, min_scale,
                                                           np.array([1.2, 1.2, 1.2]))
        if transform is None or np.linalg.det(transform[:3, :3]) < 0:
            continue
        transformed = (transform @ np.concatenate([nocs_cloud, np.ones((len(nocs_cloud), 1))], 1).T).T[:, :3]
        ratio = np.sum(np.linalg.norm(transformed - ori, axis=1) <= 0.003) / len(ori)
        if ratio > best_ratio:
            best_ratio, best_transform = ratio, transform.copy()
    return best_ratio, best_transform


def _same_outcome(out, exp, npred):
    best_ratio, best_transform = exp
    if best_transform is None:
        assert out == (None, None)
    else:
        assert out[0].shape[1] == 3 and np.abs(out[1] - best_transform).max() < 1e-9
        assert npred.best_ratio == best_ratio and np.array_equal(npred.nocs_pose, out[1])


def test_nunocs_predicter_predict_recovers_a_pose_like_the_oracle(cuda_device):
    """predict()'s post-network logic with a NUNOCS cloud that HAS an answer (random weights never give one): predict_nocs is
    replaced by a synthetic, partly wrong NUNOCS prediction of a posed object; the device RANSAC + threshold loop must return the
    oracle's transform and best_ratio exactly, and actually recover the pose."""
    from catgrasp_amd.predicter import DEFAULT_NUNOCS_CFG, NunocsPredicter
    sd = synth.make_state_dict('seg', 6, 300, seed=4)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=sd, device=cuda_device)
    src, dst, T_true = _problem(5, n=1500, outliers=0.3, noise=0.0002)
    npred.predict_nocs = lambda data, ids=None: (src.astype(np.float32), None, {'cloud_xyz_original': dst})
    np.random.seed(3)
    out = npred.predict({})
    np.random.seed(3)
    exp = _oracle_predict_tail(src.astype(np.float32), dst, npred.min_scale, npred.max_scale, 10000)
    assert exp[1] is not None and exp[0] > 0.5
    _same_outcome(out, exp, npred)
    assert np.abs(out[1][:3, 3] - T_true[:3, 3]).max() < 2e-3
