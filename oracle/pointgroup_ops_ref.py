"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy float32 restatement of the forward CUDA kernels of
PointGroup/lib/pointgroup_ops/src (bfs_cluster.cu:15-62, sec_mean.cu:12-85, roipool.cu:12-40, get_iou.cu:12-37,
voxelize.cu:10-34).  Sequential float32 accumulation in the kernels' loop order.

PARITY UNPINNED: the originals are CUDA (no nvcc / NVIDIA device here) and the reference holds no test, fixture or golden
vector for them; the semantics are restated line by line from the .cu sources."""
import numpy as np


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, radius, mean_active):
    xyz = np.asarray(xyz, dtype=np.float32)
    n = len(xyz); r2 = np.float32(radius) * np.float32(radius)
    lists = []
    for p in range(n):
        s, e = batch_offsets[batch_idxs[p]], batch_offsets[batch_idxs[p] + 1]
        d = xyz[p] - xyz[s:e]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nb = (np.flatnonzero(d2 < r2) + s)[:1000]
        lists.append(nb)
    counts = np.array([len(l) for l in lists], dtype=np.int32)
    start = (np.cumsum(counts) - counts).astype(np.int32)
    thre = n * mean_active
    idx = np.concatenate(lists)[:thre].astype(np.int32) if n else np.zeros((0,), np.int32)
    return idx, np.stack([start, counts], 1), d2


def segment(inp, offsets, mode):
    inp = np.asarray(inp, dtype=np.float32)
    nseg = len(offsets) - 1; C = inp.shape[1]
    out = np.zeros((nseg, C), dtype=np.float32); am = np.full((nseg, C), -1, dtype=np.int32)
    for s in range(nseg):
        a, b = offsets[s], offsets[s + 1]
        if mode == 0:
            acc = np.zeros(C, dtype=np.float32); cnt = np.float32(b - a)
            for i in range(a, b):
                acc = (acc + inp[i] / cnt).astype(np.float32)
            out[s] = acc
        elif mode == 1:
            out[s] = inp[a:b].min(axis=0) if b > a else np.inf
        else:
            if b > a:
                out[s] = inp[a:b].max(axis=0); am[s] = inp[a:b].argmax(axis=0) + a
            else:
                out[s] = -np.inf
    return out, am


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    nP = len(proposals_offset) - 1; nI = len(instance_pointnum)
    iou = np.zeros((nP, nI), dtype=np.float32)
    for p in range(nP):
        ids = proposals_idx[proposals_offset[p]:proposals_offset[p + 1]]
        lab = instance_labels[ids]
        for i in range(nI):
            inter = int((lab == i).sum())
            tot = len(ids) + int(instance_pointnum[i]) - inter
            iou[p, i] = np.float32(np.float64(np.float32(inter)) / (np.float64(np.float32(tot)) + 1e-5))
    return iou


def voxelize_fp(feats, rules, average):
    feats = np.asarray(feats, dtype=np.float32)
    M = len(rules); C = feats.shape[1]
    out = np.zeros((M, C), dtype=np.float32)
    for r in range(M):
        n = rules[r, 0]
        mult = np.float32(1.0) / np.float32(n) if (average and n > 0) else np.float32(1.0)
        for i in range(1, n + 1):
            out[r] = (out[r] + mult * feats[rules[r, i]]).astype(np.float32)
    return out
