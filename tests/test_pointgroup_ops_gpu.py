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
