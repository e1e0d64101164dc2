"""GPU parity of the N4 ops (PointGroup pointgroup_ops forward kernels) vs the numpy float32 restatement."""
import numpy as np
import pytest
import torch

from oracle import pointgroup_ops_ref as ref

pytestmark = pytest.mark.gpu


def test_segment_ops_and_roipool(cuda_device):
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(0)
    N, C = 700, 80                                        # C not a multiple of 64: second slab partially filled
    inp = rng.normal(size=(N, C)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, N), 12, replace=False))
    offsets = np.concatenate([[0], cuts, [N], [N]]).astype(np.int32)       # the last segment is empty
    t_in = torch.from_numpy(inp).to(cuda_device); t_off = torch.from_numpy(offsets).to(cuda_device)
    r_mean, _ = ref.segment(inp, offsets, 0); r_min, _ = ref.segment(inp, offsets, 1); r_max, r_am = ref.segment(inp, offsets, 2)
    assert np.array_equal(pg.sec_mean(t_in, t_off).cpu().numpy(), r_mean)                    # same accumulation order -> bitwise
    assert np.array_equal(pg.sec_min(t_in, t_off).cpu().numpy(), r_min)
    assert np.array_equal(pg.sec_max(t_in, t_off).cpu().numpy(), r_max)
    of, am = pg.roipool(t_in, t_off)
    assert np.array_equal(of.cpu().numpy(), r_max) and np.array_equal(am.cpu().numpy(), r_am)


def test_ballquery_batch_p(cuda_device):
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(1)
    sizes = [300, 500, 150]
    xyz = np.concatenate([rng.normal(0, 0.05, (s, 3)) + i for i, s in enumerate(sizes)]).astype(np.float32)
    batch_idxs = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)]).astype(np.int32)
    batch_offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    for mean_active in (50, 3):                           # 3 -> the n*meanActive cap truncates
        idx, start_len = pg.ballquery_batch_p(torch.from_numpy(xyz).to(cuda_device), torch.from_numpy(batch_idxs).to(cuda_device),
                                              torch.from_numpy(batch_offsets).to(cuda_device), 0.03, mean_active)
        r_idx, r_sl, _ = ref.ballquery_batch_p(xyz, batch_idxs, batch_offsets, 0.03, mean_active)
        assert np.array_equal(start_len.cpu().numpy(), r_sl)
        assert np.array_equal(idx.cpu().numpy(), r_idx)
    # neighbours never cross batches
    sl = start_len.cpu().numpy()
    assert (sl[:, 1] >= 1).all()                          # every point finds itself


def test_get_iou_and_voxelize(cuda_device):
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(2)
    N, nI, nP = 1000, 7, 9
    labels = rng.integers(-1, nI, N).astype(np.int64); labels[labels < 0] = -100
    pointnum = np.array([(labels == i).sum() for i in range(nI)], dtype=np.int32)
    lens = rng.integers(1, 200, nP); off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    pidx = rng.integers(0, N, off[-1]).astype(np.int32)
    got = pg.get_iou(torch.from_numpy(pidx).to(cuda_device), torch.from_numpy(off).to(cuda_device), torch.from_numpy(labels).to(cuda_device),
                     torch.from_numpy(pointnum).to(cuda_device)).cpu().numpy()
    assert np.array_equal(got, ref.get_iou(pidx, off, labels, pointnum))
    C, M, maxA = 70, 40, 6
    feats = rng.normal(size=(N, C)).astype(np.float32)
    rules = np.zeros((M, maxA + 1), dtype=np.int32)
    for r in range(M):
        k = rng.integers(0, maxA + 1); rules[r, 0] = k; rules[r, 1:1 + k] = rng.integers(0, N, k)
    for mode, avg in ((4, True), (3, False)):
        v = pg.voxelization(torch.from_numpy(feats).to(cuda_device), torch.from_numpy(rules).to(cuda_device), mode).cpu().numpy()
        assert np.array_equal(v, ref.voxelize_fp(feats, rules, avg))


def test_voxelization_idx_matches_the_host_rulebook(cuda_device):
    """voxelize.cpp builds the maps with an insertion-ordered std::map on the host; the device sort/scan formulation must return
    the identical three tensors (first-appearance voxel order, ascending members), for every mode and both coordinate layouts."""
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(3)
    n = 5000
    coords4 = np.concatenate([rng.integers(0, 3, (n, 1)), rng.integers(0, 12, (n, 3))], axis=1).astype(np.int64)   # many duplicates
    coords4 = coords4[np.argsort(coords4[:, 0], kind='stable')]          # batches contiguous like the real loader, order inside random
    for coords, modes in ((coords4, (4, 3, 1, 2)), (coords4[:, 1:], (4,))):
        for mode in modes:
            oc, im, om = pg.voxelization_idx(torch.from_numpy(coords).to(cuda_device), 3, mode)
            roc, rim, rom = ref.voxelization_idx(coords, mode)
            assert np.array_equal(oc.cpu().numpy(), roc) and np.array_equal(im.cpu().numpy(), rim) and np.array_equal(om.cpu().numpy(), rom)
    uniq = np.unique(coords4, axis=0)
    rng.shuffle(uniq)
    oc, im, om = pg.voxelization_idx(torch.from_numpy(uniq).to(cuda_device), 3, 0)
    assert np.array_equal(im.cpu().numpy(), np.arange(len(uniq))) and np.array_equal(oc.cpu().numpy(), uniq)
    # the rule book drives the pooling kernel exactly like the reference's (predicter.py:285-286)
    feats = rng.normal(size=(n, 7)).astype(np.float32)
    _, _, om = pg.voxelization_idx(torch.from_numpy(coords4).to(cuda_device), 3, 4)
    pooled = pg.voxelization(torch.from_numpy(feats).to(cuda_device), om, 4).cpu().numpy()
    assert np.array_equal(pooled, ref.voxelize_fp(feats, ref.voxelization_idx(coords4, 4)[2], True))
    with pytest.raises(ValueError):
        pg.voxelization_idx(torch.tensor([[0, 1, 2, 70000]], device=cuda_device), 1, 4)


def test_bfs_cluster_matches_the_host_bfs(cuda_device):
    """bfs_cluster.cpp's sequential queue BFS vs min-label propagation on the device: same clusters (as point sets), same
    numbering (by smallest point index), same offsets; chains (worst case for propagation), isolated points, label boundaries."""
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(5)
    blobs = [rng.normal(c, 0.02, (m, 3)) for c, m in (((0, 0, 0), 400), ((0.3, 0, 0), 300), ((0, 0.3, 0), 90), ((1, 1, 1), 30))]
    chain = np.stack([np.linspace(2, 3.5, 300), np.zeros(300), np.zeros(300)], 1)          # a 300-point path: diameter 299
    xyz = np.concatenate(blobs + [chain, rng.uniform(5, 9, (40, 3))]).astype(np.float32)
    perm = rng.permutation(len(xyz)); xyz = xyz[perm]
    label = rng.integers(0, 2, len(xyz)).astype(np.int32)                                  # two classes interleaved in space
    n = len(xyz)
    bi = np.zeros(n, dtype=np.int32); bo = np.array([0, n], dtype=np.int32)
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    idx, start_len = pg.ballquery_batch_p(t(xyz), t(bi), t(bo), 0.03, 300)
    for thr in (1, 50):
        ci, co_ = pg.bfs_cluster(t(label), idx, start_len, thr)
        rci, rco = ref.bfs_cluster(label, idx.cpu().numpy(), start_len.cpu().numpy(), thr)
        ci, co_ = ci.cpu().numpy(), co_.cpu().numpy()
        assert np.array_equal(co_, rco) and np.array_equal(ci[:, 0], rci[:, 0])
        for c in range(len(rco) - 1):
            assert np.array_equal(ci[rco[c]:rco[c + 1], 1], np.sort(rci[rco[c]:rco[c + 1], 1]))
    assert len(rco) - 1 >= 3


def test_host_side_ops_against_the_reference_cpp_golden(cuda_device):
    """The device formulations of voxelization_idx / bfs_cluster vs outputs of the REFERENCE's own host C++ (pointgroup_golden.npz):
    the three voxelization tensors identical for every mode; the same clusters, numbering and offsets (members as sets: the
    reference lists them in queue-visit order, the device in ascending index)."""
    import os
    from catgrasp_amd import pointgroup_ops as pg
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pointgroup_golden.npz'))
    for mode, coords in ((4, g['vox_coords']), (3, g['vox_coords']), (1, g['vox_coords']), (2, g['vox_coords']), (0, g['vox_unique_coords'])):
        oc, im, om = pg.voxelization_idx(torch.from_numpy(coords).to(cuda_device), 3, mode)
        assert np.array_equal(oc.cpu().numpy(), g[f'vox_mode{mode}_output_coords']) and np.array_equal(im.cpu().numpy(), g[f'vox_mode{mode}_input_map'])
        assert np.array_equal(om.cpu().numpy(), g[f'vox_mode{mode}_output_map'])
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    for thr in (1, 50):
        ci, co_ = pg.bfs_cluster(t(g['bfs_label']), t(g['bfs_idx']), t(g['bfs_start_len']), thr)
        rci, rco = g[f'bfs_thr{thr}_cluster_idxs'], g[f'bfs_thr{thr}_cluster_offsets']
        ci, co_ = ci.cpu().numpy(), co_.cpu().numpy()
        assert np.array_equal(co_, rco) and np.array_equal(ci[:, 0], rci[:, 0])
        for c in range(len(rco) - 1):
            assert np.array_equal(ci[rco[c]:rco[c + 1], 1], np.sort(rci[rco[c]:rco[c + 1], 1]))
