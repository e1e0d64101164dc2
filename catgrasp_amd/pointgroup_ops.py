"""Row N4: forward passes of the reference's `pointgroup_ops` CUDA extension
(PointGroup/lib/pointgroup_ops/functions/pointgroup_ops.py) used on the inference path (predicter.py:285-304), on HIP.
Same function names and tensor conventions (int32 index tensors, CSR offsets)."""
import ctypes

import torch

from . import _lib as L
from ._lib import _p, _stream, check, require_cuda

_c_int = ctypes.c_int


def _on_device(*tensors):
    """The reference calls voxelization_idx / bfs_cluster with CPU tensors (predicter.py:285: `locs` is a CPU LongTensor;
    pointgroup.py:240,245: semantic_preds_cpu, idx.cpu(), start_len.cpu()) and indexes CPU tensors with the results.  Host tensors are
    therefore accepted: they are uploaded to the current HIP device, the kernels run there (there is no CPU implementation), and the
    results go back to the inputs' device.  -> (device tensors ..., device of the first input)."""
    home = tensors[0].device
    if all(t.is_cuda for t in tensors):
        return tensors + (home,)
    if not torch.cuda.is_available():
        raise L.CatgraspAmdError('catgrasp_amd.pointgroup_ops needs a HIP device (host tensors are uploaded to it; there is no CPU fallback)')
    dev = next((t.device for t in tensors if t.is_cuda), torch.device('cuda', torch.cuda.current_device()))
    return tuple(t.to(dev) for t in tensors) + (home,)


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """-> (idx (nActive,) int32, start_len (n,2) int32): neighbours of every point within `radius` inside its own batch
    (pointgroup_ops.py BallQueryBatchP; bfs_cluster.cu:15-91).  The reference hands out CSR start positions with an
    atomicAdd (order = thread arrival); here they are the prefix sum in point order, one of its valid outcomes, with the
    same 1000-neighbour cap per point.  `meanActive` is only the reference's first guess of the list capacity: its wrapper retries
    with a larger one until nActive <= n*meanActive (pointgroup_ops.py:134-141), so the lists it returns are never cut -- the counts
    of the first pass size `idx` exactly here, which is that loop's fixed point."""
    require_cuda(coords, batch_idxs, batch_offsets)
    coords = coords.contiguous().float(); batch_idxs = batch_idxs.contiguous().int(); batch_offsets = batch_offsets.contiguous().int()
    n = coords.shape[0]
    counts = torch.zeros((n,), dtype=torch.int32, device=coords.device)
    lib = L.lib()
    check(lib.cg_pg_ballquery_batch_p(_p(coords), _p(batch_idxs), _p(batch_offsets), _c_int(n), ctypes.c_float(radius), _c_int(0), None, None,
                                      _p(counts), None, _stream()), 'cg_pg_ballquery_batch_p')
    csum = torch.cumsum(counts.long(), 0)
    total = int(csum[-1].item()) if n else 0
    if total >= 2 ** 31:
        raise ValueError('ballquery_batch_p: more than 2^31 neighbour entries')
    start = (csum - counts).int().contiguous()
    idx = torch.zeros((max(total, 1),), dtype=torch.int32, device=coords.device)
    check(lib.cg_pg_ballquery_batch_p(_p(coords), _p(batch_idxs), _p(batch_offsets), _c_int(n), ctypes.c_float(radius), _c_int(1), _p(start),
                                      _p(counts), None, _p(idx), _stream()), 'cg_pg_ballquery_batch_p')
    start_len = torch.stack([start, counts], dim=1).contiguous()
    return idx[:total].contiguous(), start_len


def _segment(inp, offsets, mode, want_argmax=False):
    require_cuda(inp, offsets)
    inp = inp.contiguous().float(); offsets = offsets.contiguous().int()
    nseg = offsets.shape[0] - 1
    C = inp.shape[1]
    out = torch.zeros((nseg, C), dtype=torch.float32, device=inp.device)
    am = torch.zeros((nseg, C), dtype=torch.int32, device=inp.device) if want_argmax else None
    check(L.lib().cg_pg_segment_reduce(_p(inp), _p(offsets), _c_int(nseg), _c_int(C), _c_int(mode), _p(out), _p(am), _stream()),
          'cg_pg_segment_reduce')
    return (out, am) if want_argmax else out


def sec_mean(inp, offsets):
    return _segment(inp, offsets, 0)


def sec_min(inp, offsets):
    return _segment(inp, offsets, 1)


def sec_max(inp, offsets):
    return _segment(inp, offsets, 2)


def roipool(feats, proposals_offset):
    """RoiPool forward: (output_feats (nProposal,C), output_maxidx (nProposal,C) int32)."""
    return _segment(feats, proposals_offset, 3, want_argmax=True)


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    require_cuda(proposals_idx, proposals_offset, instance_labels, instance_pointnum)
    nI = instance_pointnum.shape[0]; nP = proposals_offset.shape[0] - 1
    iou = torch.zeros((nP, nI), dtype=torch.float32, device=proposals_idx.device)
    check(L.lib().cg_pg_get_iou(_p(proposals_idx.contiguous().int()), _p(proposals_offset.contiguous().int()), _p(instance_labels.contiguous().long()),
                                _p(instance_pointnum.contiguous().int()), _c_int(nP), _c_int(nI), _p(iou), _stream()), 'cg_pg_get_iou')
    return iou


def voxelization(feats, map_rule, mode=4):
    """Voxelization forward (pointgroup_ops.py Voxelization): map_rule (M, 1+maxActive) int32; mode 4 = mean, else sum."""
    require_cuda(feats, map_rule)
    feats = feats.contiguous().float(); rules = map_rule.contiguous().int()
    M, width = rules.shape
    C = feats.shape[1]
    out = torch.zeros((M, C), dtype=torch.float32, device=feats.device)
    check(L.lib().cg_pg_voxelize_fp(_p(feats), _p(rules), _c_int(M), _c_int(width - 1), _c_int(C), _c_int(int(mode == 4)), _p(out), _stream()),
          'cg_pg_voxelize_fp')
    return out


def point_recover(feats, map_rule, nPoint):
    """PointRecover forward (pointgroup_ops.py:77-99): feats (M,C) voxel features, map_rule (M, 1+maxActive) int32 [count, point ids...]
    -> (nPoint, C): every voxel's row added onto each of its member points (zeros for points no voxel lists).  A count above
    maxActive or a member id outside [0, nPoint) raises ValueError (the reference writes out of bounds)."""
    require_cuda(feats, map_rule)
    feats = feats.contiguous().float(); rules = map_rule.contiguous().int()
    M, width = rules.shape
    if feats.dim() != 2 or feats.shape[0] != M:
        raise ValueError(f'feats {tuple(feats.shape)} does not have one row per rule ({M})')
    C = feats.shape[1]
    out = torch.zeros((int(nPoint), C), dtype=torch.float32, device=feats.device)
    err = torch.zeros((1,), dtype=torch.int32, device=feats.device)
    check(L.lib().cg_pg_point_recover(_p(feats), _p(rules), _c_int(M), _c_int(width - 1), _c_int(C), _c_int(int(nPoint)), _p(out), _p(err),
                                      _stream()), 'cg_pg_point_recover')
    if int(err.item()):
        raise ValueError('point_recover: a rule lists more members than maxActive or a point outside [0, nPoint)')
    return out


def voxelization_idx(coords, batchsize, mode=4):
    """pointgroup_ops.voxelization_idx (Voxelization_Idx.forward; voxelize.cpp:11-151; called at predicter.py:285):
    coords (N,4) [batch,x,y,z] or (N,3) int64 -> (output_coords (M,ncol) int64, input_map (N) int32, output_map (M, 1+maxActive)
    int32).  Voxels are numbered in order of first appearance; output_map row = [count, member point ids ascending, 0 ...]
    (mode 3/4), [1, first] (mode 1), [1, last] (mode 2).  The reference builds the maps on the host from CPU tensors
    (predicter.py:285 passes a CPU LongTensor and gets CPU tensors back); the maps are built by device kernels here, and the
    returned tensors live on coords.device -- a host `coords` is uploaded and its results are downloaded (see _on_device)."""
    coords, home = _on_device(coords)
    if home != coords.device:
        return tuple(t.to(home) for t in voxelization_idx(coords, batchsize, mode))
    coords = coords.contiguous().long()
    n, ncol = coords.shape
    dev = coords.device
    if n == 0:
        return coords.new_zeros((0, ncol)), torch.zeros((0,), dtype=torch.int32, device=dev), torch.zeros((0, 2), dtype=torch.int32, device=dev)
    lib = L.lib()
    keys = torch.empty((n,), dtype=torch.int64, device=dev)
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    check(lib.cg_pg_voxel_pack_keys(_p(coords), _c_int(n), _c_int(ncol), _p(keys), _p(err), _stream()), 'cg_pg_voxel_pack_keys')
    skeys, perm = torch.sort(keys, stable=True)                         # ties keep point order
    head = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib.cg_pg_segment_heads(_p(skeys), _c_int(n), _p(head), _stream()), 'cg_pg_segment_heads')
    seg = (torch.cumsum(head, 0) - 1).int()
    seg_start = torch.nonzero(head, as_tuple=False).reshape(-1).int()  # also: its length M and the sync point
    if int(err.item()):
        raise ValueError('voxelization_idx: coordinates must lie in [0, 65536) and batch indices in [0, 32768)')
    M = seg_start.shape[0]
    first_pt = perm[seg_start.long()]                                   # smallest point index of every run (stable sort)
    order = torch.argsort(first_pt)                                     # runs in first-appearance order
    vid = torch.empty((M,), dtype=torch.int32, device=dev)
    vid[order] = torch.arange(M, dtype=torch.int32, device=dev)
    if mode in (3, 4):
        counts = torch.diff(torch.cat([seg_start, torch.tensor([n], dtype=torch.int32, device=dev)]))
        max_active = max(int(counts.max().item()), 1)
    else:
        max_active = 1
        if mode == 0 and M != n:
            raise ValueError('voxelization_idx mode 0 needs unique coordinates')
    input_map = torch.empty((n,), dtype=torch.int32, device=dev)
    output_map = torch.zeros((M, max_active + 1), dtype=torch.int32, device=dev)
    check(lib.cg_pg_voxel_fill_maps(_p(perm), _p(seg), _p(seg_start), _p(vid), _c_int(n), _c_int(max_active + 1), _c_int(int(mode)),
                                    _p(input_map), _p(output_map), _stream()), 'cg_pg_voxel_fill_maps')
    output_coords = coords[output_map[:, 1].long()]                     # voxelize_outputmap: coords of rule[1]
    return output_coords, input_map, output_map


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """pointgroup_ops.bfs_cluster (BFSCluster.forward; bfs_cluster.cpp:34-121; called at pointgroup.py:240,245):
    -> (cluster_idxs (sumNPoint,2) int32 [cluster id, point id], cluster_offsets (nCluster+1) int32).  Clusters = connected
    components of the neighbour graph restricted to equal semantic labels with >= threshold points, numbered by their smallest
    point index (the order the reference's seed loop finds them).  Members are listed in ascending point index; the reference
    lists them in queue-visit order (same sets).  The neighbour relation is used symmetrically (it is symmetric unless the ball
    query's 1000-neighbour cap truncated a list).  Host tensors are accepted like the reference's (pointgroup.py:240,245 pass
    `.cpu()` tensors) and the results returned on the inputs' device (see _on_device).  A CSR row that reaches past the end of
    `ball_query_idxs`, or an entry outside [0, N), raises (the reference's queue BFS would read out of bounds there)."""
    semantic_label, ball_query_idxs, start_len, home = _on_device(semantic_label, ball_query_idxs, start_len)
    if home != semantic_label.device:
        return tuple(t.to(home) for t in bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold))
    label = semantic_label.contiguous().int(); nbr = ball_query_idxs.contiguous().int(); sl = start_len.contiguous().int()
    n = sl.shape[0]
    dev = label.device
    if label.shape[0] != n:
        raise ValueError('bfs_cluster: semantic_label and start_len disagree on the number of points')
    n_idx = nbr.shape[0]
    if n:
        ends = sl[:, 0].long() + sl[:, 1].long()
        if int(sl.min().item()) < 0 or int(ends.max().item()) > n_idx:
            raise ValueError(f'bfs_cluster: start_len addresses entries beyond the {n_idx} of ball_query_idxs (a truncated neighbour list?)')
        if n_idx and (int(nbr.min().item()) < 0 or int(nbr.max().item()) >= n):
            raise ValueError('bfs_cluster: ball_query_idxs holds a point index outside [0, N)')
    comp = torch.arange(n, dtype=torch.int32, device=dev)
    changed = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = L.lib()
    for _ in range(max(n, 1) + 1):
        changed.zero_()
        for _ in range(4):                                              # a few sweeps per host round trip
            check(lib.cg_pg_cc_propagate(_p(label), _p(nbr), _c_int(n_idx), _p(sl), _c_int(n), _p(comp), _p(changed), _stream()), 'cg_pg_cc_propagate')
        if int(changed.item()) == 0:
            break
    sizes = torch.bincount(comp.long(), minlength=n)
    keep = sizes[comp.long()] >= threshold
    root, pts = comp[keep], torch.arange(n, dtype=torch.int32, device=dev)[keep]
    order = torch.sort(root.long() * n + pts.long()).indices          # by (root, point id)
    root, pts = root[order], pts[order]
    uniq, inverse, counts = torch.unique_consecutive(root, return_inverse=True, return_counts=True)
    cluster_idxs = torch.stack([inverse.int(), pts], dim=1).contiguous()
    offsets = torch.zeros((uniq.shape[0] + 1,), dtype=torch.int32, device=dev)
    offsets[1:] = torch.cumsum(counts, 0).int()
    return cluster_idxs, offsets
