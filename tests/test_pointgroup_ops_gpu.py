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
    for mean_active in (50, 3):                           # 3 -> the reference wrapper's first guess is too small and it retries: never truncated
        idx, start_len = pg.ballquery_batch_p(torch.from_numpy(xyz).to(cuda_device), torch.from_numpy(batch_idxs).to(cuda_device),
                                              torch.from_numpy(batch_offsets).to(cuda_device), 0.03, mean_active)
        r_idx, r_sl, _ = ref.ballquery_batch_p(xyz, batch_idxs, batch_offsets, 0.03, mean_active)
        assert np.array_equal(start_len.cpu().numpy(), r_sl)
        assert np.array_equal(idx.cpu().numpy(), r_idx)
        assert idx.shape[0] == int(r_sl[:, 1].sum()) > len(xyz) * 3      # the full lists, beyond n * 3
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


def test_point_recover_puts_voxel_features_back_on_their_points(cuda_device):
    """point_recover (pointgroup_ops.py:77-99; voxelize.cpp:182-192): against the oracle on a rule book made by voxelization_idx
    (every point gets exactly its voxel's row: bit-exact), on rule books with empty rows, rows listing fewer members than maxActive and a
    point listed by two rows (a sum), for C below and above one wavefront; bad rule books raise instead of writing out of bounds."""
    from catgrasp_amd import pointgroup_ops as pg
    rng = np.random.default_rng(11)
    n = 4000
    coords = np.concatenate([rng.integers(0, 2, (n, 1)), rng.integers(0, 9, (n, 3))], axis=1).astype(np.int64)
    coords = coords[np.argsort(coords[:, 0], kind='stable')]
    _, im, om = pg.voxelization_idx(torch.from_numpy(coords).to(cuda_device), 2, 4)
    for C in (1, 16, 33, 150):
        pt = rng.normal(size=(n, C)).astype(np.float32)
        vox = pg.voxelization(torch.from_numpy(pt).to(cuda_device), om, 4)
        back = pg.point_recover(vox, om, n)
        assert back.shape == (n, C) and np.array_equal(back.cpu().numpy(), ref.point_recover(vox.cpu().numpy(), om.cpu().numpy(), n))
        assert torch.equal(back, vox[im.long()])                         # == the gather through the input map
    rules = np.zeros((6, 5), dtype=np.int32)
    rules[0] = [3, 0, 1, 2, 99]; rules[1] = [0, 7, 7, 7, 7]; rules[2] = [1, 4, 0, 0, 0]; rules[3] = [4, 5, 6, 7, 8]; rules[4] = [2, 2, 9, 0, 0]      # point 2 twice
    feats = rng.normal(size=(6, 5)).astype(np.float32)
    got = pg.point_recover(torch.from_numpy(feats).to(cuda_device), torch.from_numpy(rules).to(cuda_device), 12).cpu().numpy()
    want = ref.point_recover(feats, rules, 12)
    assert np.array_equal(got[[0, 1, 4, 5, 6, 7, 8, 9, 10, 11]], want[[0, 1, 4, 5, 6, 7, 8, 9, 10, 11]]) and np.allclose(got[2], want[2], atol=1e-6)
    assert not got[3].any() and not got[10:].any()
    for bad in ([2, 0, 12, 0, 0], [5, 0, 1, 2, 3], [1, -1, 0, 0, 0]):
        r2 = rules.copy(); r2[5] = bad
        with pytest.raises(ValueError):
            pg.point_recover(torch.from_numpy(feats).to(cuda_device), torch.from_numpy(r2).to(cuda_device), 12)
    assert pg.point_recover(torch.zeros((0, 4), device=cuda_device), torch.zeros((0, 3), dtype=torch.int32, device=cuda_device), 5).shape == (5, 4)


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
    # the reference call sites hand CPU tensors over and index CPU tensors with the result (pointgroup.py:240,245; predicter.py:285):
    # host tensors are accepted, the kernels still run on the device, the results come back on the host
    ci_h, co_h = pg.bfs_cluster(torch.from_numpy(label), idx.cpu(), start_len.cpu(), 50)
    assert not ci_h.is_cuda and not co_h.is_cuda and np.array_equal(ci_h.numpy(), ci) and np.array_equal(co_h.numpy(), co_)
    coords = torch.from_numpy(np.concatenate([np.zeros((n, 1)), np.floor(xyz * 50) + 10], 1).astype(np.int64))
    oc_h, im_h, om_h = pg.voxelization_idx(coords, 1, 4)
    oc_d, im_d, om_d = pg.voxelization_idx(coords.to(cuda_device), 1, 4)
    assert not oc_h.is_cuda and not im_h.is_cuda and not om_h.is_cuda
    assert torch.equal(oc_h, oc_d.cpu()) and torch.equal(im_h, im_d.cpu()) and torch.equal(om_h, om_d.cpu())
    # a CSR row that reaches past the end of the neighbour array (a truncated list) is rejected, not read
    with pytest.raises(ValueError):
        pg.bfs_cluster(t(label), idx[:idx.shape[0] // 2].contiguous(), start_len, 1)
    bad = idx.clone(); bad[5] = n + 7
    with pytest.raises(ValueError):
        pg.bfs_cluster(t(label), bad, start_len, 1)
    # ... and the kernel itself is memory-safe on such input (rows clamped to n_idx, foreign indices skipped): call the C ABI directly
    import ctypes
    from catgrasp_amd import _lib as L
    comp = torch.arange(n, dtype=torch.int32, device=cuda_device); changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    half = bad[:idx.shape[0] // 2].contiguous()
    st = L.lib().cg_pg_cc_propagate(L._p(t(label)), L._p(half), ctypes.c_int(half.shape[0]), L._p(start_len.contiguous()), ctypes.c_int(n), L._p(comp),
                                    L._p(changed), L._stream())
    torch.cuda.synchronize()
    assert st == 0 and int(comp.min()) >= 0 and int(comp.max()) < n


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


def test_product_equals_the_reference_cuda_kernels_running_on_this_gpu(cuda_device):
    """Row N4 pinned to the reference implementation itself: oracle/_ref/libpointgroup_kernels_ref.so holds the reference's OWN
    PointGroup CUDA kernels (bfs_cluster.cu:15-62, sec_mean.cu, roipool.cu:12-31, get_iou.cu:12-29, voxelize.cu:9-23,34-48), compiled for
    gfx950 by oracle/build_ref.py:build_pointgroup_kernels from the text where it lies and launched with the reference's geometry.
    The product kernels must return the same tensors on the same device inputs: bitwise for the segmented reductions, the arg-max
    pool, the IoU table and the rule-book pooling; per-point neighbour lists and counts for the ball query (the reference hands out
    its CSR start positions with an atomicAdd, i.e. in thread-arrival order)."""
    import ctypes
    import os
    from catgrasp_amd import pointgroup_ops as pg
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libpointgroup_kernels_ref.so')
    if not os.path.exists(so):
        pytest.skip('oracle/_ref/libpointgroup_kernels_ref.so not built (python oracle/build_ref.py in the build container)')
    lib = ctypes.CDLL(so)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = cuda_device
    g = torch.Generator(device=dev); g.manual_seed(3)
    # ---- segmented reductions + RoI pool
    N, C = 3000, 80
    inp = torch.randn(N, C, device=dev, generator=g)
    cuts = torch.sort(torch.randperm(N - 1, device=dev, generator=g)[:40] + 1).values
    offsets = torch.cat([torch.zeros(1, device=dev, dtype=torch.long), cuts, torch.tensor([N], device=dev)]).int().contiguous()
    nseg = offsets.numel() - 1
    for which, fn in ((0, pg.sec_mean), (1, pg.sec_min), (2, pg.sec_max)):
        ref_out = torch.zeros(nseg, C, device=dev)
        assert lib.ref_sec(which, nseg, C, P(inp), P(offsets), P(ref_out), st) == 0
        assert torch.equal(fn(inp, offsets), ref_out)
    ref_feats = torch.zeros(nseg, C, device=dev); ref_idx = torch.zeros(nseg, C, device=dev, dtype=torch.int32)
    assert lib.ref_roipool_fp(nseg, C, P(inp), P(offsets), P(ref_feats), P(ref_idx), st) == 0
    of, am = pg.roipool(inp, offsets)
    assert torch.equal(of, ref_feats) and torch.equal(am, ref_idx)
    # ---- IoU of proposals vs instances
    nI = 37
    labels = torch.randint(-1, nI, (N,), device=dev, generator=g).long(); labels[labels < 0] = -100
    pidx = torch.randint(0, N, (5000,), device=dev, generator=g).int()
    poff = torch.tensor([0, 100, 100, 900, 2500, 5000], device=dev, dtype=torch.int32)
    pnum = torch.bincount(labels[labels >= 0], minlength=nI).int()
    ref_iou = torch.zeros(5, nI, device=dev)
    assert lib.ref_get_iou(nI, 5, P(pidx), P(poff), P(labels), P(pnum), P(ref_iou), st) == 0
    assert torch.equal(pg.get_iou(pidx, poff, labels, pnum), ref_iou)
    # ---- rule-book pooling (mean and sum) through the product's own voxelization_idx maps
    coords = torch.cat([torch.zeros(N, 1, dtype=torch.long, device=dev), torch.randint(0, 9, (N, 3), device=dev, generator=g)], 1)
    _, _, om = pg.voxelization_idx(coords, 1, 4)
    feats = torch.randn(N, 19, device=dev, generator=g)
    for mode in (4, 3):
        ref_v = torch.zeros(om.shape[0], 19, device=dev)
        assert lib.ref_voxelize_fp(om.shape[0], om.shape[1] - 1, 19, P(feats), P(ref_v), P(om), int(mode == 4), st) == 0
        assert torch.equal(pg.voxelization(feats, om, mode), ref_v)
    # ---- point_recover: the reference's voxelize backward kernel on (voxel features, zeroed point features) (voxelize.cpp:182-192)
    if hasattr(lib, 'ref_point_recover_fp'):
        vox = pg.voxelization(feats, om, 4).contiguous()
        ref_p = torch.zeros(N, 19, device=dev)
        assert lib.ref_point_recover_fp(om.shape[0], om.shape[1] - 1, 19, P(vox), P(ref_p), P(om), st) == 0
        assert torch.equal(pg.point_recover(vox, om, N), ref_p)
    # ---- ball query inside batches
    sizes = [700, 1200, 300]
    xyz = torch.cat([torch.rand(s, 3, device=dev, generator=g) * 0.3 for s in sizes]).contiguous()
    bidx = torch.cat([torch.full((s,), i, dtype=torch.int32, device=dev) for i, s in enumerate(sizes)])
    boff = torch.tensor([0, 700, 1900, 2200], dtype=torch.int32, device=dev)
    n, mean_active, radius = xyz.shape[0], 60, 0.04
    ref_idx = torch.zeros(n * mean_active, dtype=torch.int32, device=dev); ref_sl = torch.zeros(n, 2, dtype=torch.int32, device=dev)
    cumsum = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.ref_ballquery_batch_p(n, mean_active, ctypes.c_float(radius), P(xyz), P(bidx), P(boff), P(ref_idx), P(ref_sl), P(cumsum), st) == 0
    idx, sl = pg.ballquery_batch_p(xyz, bidx, boff, radius, mean_active)
    total = int(cumsum.item())
    assert total == idx.numel() < n * mean_active and torch.equal(sl[:, 1], ref_sl[:, 1])
    ri, rs, mi, ms = ref_idx.cpu().numpy(), ref_sl.cpu().numpy(), idx.cpu().numpy(), sl.cpu().numpy()
    # the reference hands out start offsets by atomicAdd in thread-arrival order, the product by a prefix sum in point order: both tile
    # [0, total) without gaps, but the start offsets themselves differ -- the per-point lists are what must agree
    for st_len in (rs, ms):
        o = np.argsort(st_len[:, 0], kind='stable')
        nz = st_len[o][st_len[o][:, 1] > 0]
        assert nz[0, 0] == 0 and np.array_equal(nz[1:, 0], (nz[:, 0] + nz[:, 1])[:-1]) and nz[-1, 0] + nz[-1, 1] == total
    for p in range(n):
        assert np.array_equal(ri[rs[p, 0]:rs[p, 0] + rs[p, 1]], mi[ms[p, 0]:ms[p, 0] + ms[p, 1]])
