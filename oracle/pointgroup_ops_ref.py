"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy float32 restatement of the forward CUDA kernels of
PointGroup/lib/pointgroup_ops/src (bfs_cluster.cu:15-62, sec_mean.cu:12-85, roipool.cu:12-40, get_iou.cu:12-37,
voxelize.cu:10-34).  Sequential float32 accumulation in the kernels' loop order.

PARITY: the reference holds no test, fixture or golden vector for its CUDA kernels (ballquery_batch_p, sec_*, roipool, get_iou,
voxelize_fp) and there is no nvcc / NVIDIA device here; this file restates their semantics line by line from the .cu sources.  The
PRODUCT kernels are additionally pinned to the reference kernels THEMSELVES: they are plain CUDA C, so oracle/build_ref.py:
build_pointgroup_kernels compiles their text (from where it lies) for gfx950 with hipcc, and
tests/test_pointgroup_ops_gpu.py::test_product_equals_the_reference_cuda_kernels_running_on_this_gpu runs both on the MI355X.
The two HOST-side ops at the end of this file (voxelization_idx, bfs_cluster) ARE pinned to the reference's own C++: oracle/build_ref.py:
build_pointgroup_host compiles voxelize.cpp:34-152 and bfs_cluster.cpp:33-91 from the lines where they lie (+ datatype.cpp; the
absent google-sparsehash container replaced by the stand-in oracle/pg_shim), tests/golden/make_golden_pointgroup.py commits its
outputs (tests/golden/pointgroup_golden.npz) and tests/test_oracle_host_golden.py requires these restatements to reproduce them exactly."""
import numpy as np


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, radius, mean_active):
    """BallQueryBatchP.forward (pointgroup_ops.py:114-143) around the kernel of bfs_cluster.cu:15-62: the kernel cuts its writes at
    n*meanActive but always returns the full count nActive; the wrapper retries with meanActive = nActive // n + 1 until everything
    fits, so the lists it hands back are never truncated (only the 1000-neighbour cap per point remains)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = len(xyz); r2 = np.float32(radius) * np.float32(radius)
    lists = []
    d2 = None
    for p in range(n):
        s, e = batch_offsets[batch_idxs[p]], batch_offsets[batch_idxs[p] + 1]
        d = xyz[p] - xyz[s:e]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nb = (np.flatnonzero(d2 < r2) + s)[:1000]
        lists.append(nb)
    counts = np.array([len(l) for l in lists], dtype=np.int32)
    start = (np.cumsum(counts) - counts).astype(np.int32)
    n_active = int(counts.sum())
    while True:                                     # pointgroup_ops.py:134-141
        thre = n * mean_active
        idx = np.concatenate(lists)[:thre].astype(np.int32) if n else np.zeros((0,), np.int32)      # the kernel's cut (bfs_cluster.cu:52-57)
        if n_active <= thre:
            break
        mean_active = int(n_active // n + 1)
    return idx[:n_active], np.stack([start, counts], 1), d2


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


def point_recover(feats, rules, n_point):
    """voxelize.cpp:182-192 (point_recover_fp = voxelize_bp_cuda_ with average = false, voxelize.cu:34-48): row m of `feats` is added to
    every member point rules[m][1..count]."""
    feats = np.asarray(feats, dtype=np.float32)
    out = np.zeros((n_point, feats.shape[1]), dtype=np.float32)
    for r in range(len(rules)):
        for i in range(1, rules[r, 0] + 1):
            out[rules[r, i]] = (out[rules[r, i]] + feats[r]).astype(np.float32)
    return out


def voxelization_idx(coords, mode=4):
    """voxelize.cpp:58-151 (voxelize_inputmap + voxelize_outputmap): an insertion-ordered map from coordinate to voxel id."""
    coords = np.asarray(coords, dtype=np.int64)
    table, rows = {}, []
    input_map = np.zeros((len(coords),), dtype=np.int32)
    for i, c in enumerate(coords):
        k = tuple(c.tolist())
        if k not in table:
            table[k] = len(rows); rows.append([])
        rows[table[k]].append(i)
        input_map[i] = table[k]
    if mode in (3, 4):
        max_active = max([len(r) for r in rows] + [1])
        out_map = np.zeros((len(rows), max_active + 1), dtype=np.int32)
        for v, r in enumerate(rows):
            out_map[v, 0] = len(r); out_map[v, 1:1 + len(r)] = r
    else:
        out_map = np.zeros((len(rows), 2), dtype=np.int32)
        for v, r in enumerate(rows):
            out_map[v] = [1, r[0] if mode in (0, 1) else r[-1]]
    out_coords = coords[out_map[:, 1]] if len(rows) else np.zeros((0, coords.shape[1]), dtype=np.int64)
    return out_coords, input_map, out_map


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """bfs_cluster.cpp:34-121: seed loop over unvisited points in index order, queue BFS over same-label neighbours;
    -> (cluster_idxs (sumNPoint,2) in visit order, cluster_offsets)."""
    from collections import deque
    n = len(start_len)
    visited = np.zeros(n, dtype=bool)
    clusters = []
    for i in range(n):
        if visited[i]:
            continue
        cc = [i]; visited[i] = True
        q = deque([i])
        while q:
            cur = q.popleft()
            s, l = start_len[cur]
            for j in ball_query_idxs[s:s + l]:
                if semantic_label[j] != semantic_label[cur] or visited[j]:
                    continue
                cc.append(int(j)); visited[j] = True; q.append(int(j))
        if len(cc) >= threshold:
            clusters.append(cc)
    idxs = np.array([[c, p] for c, cc in enumerate(clusters) for p in cc], dtype=np.int32).reshape(-1, 2)
    offsets = np.concatenate([[0], np.cumsum([len(cc) for cc in clusters])]).astype(np.int32)
    return idxs, offsets
