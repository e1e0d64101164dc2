"""Row N4: forward passes of the reference's `pointgroup_ops` CUDA extension
(PointGroup/lib/pointgroup_ops/functions/pointgroup_ops.py) used on the inference path (predicter.py:285-304), on HIP.
Same function names and tensor conventions (int32 index tensors, CSR offsets)."""
import ctypes

import torch

from . import _lib as L
from ._lib import _p, _stream, check, require_cuda

_c_int = ctypes.c_int


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """-> (idx (nActive,) int32, start_len (n,2) int32): neighbours of every point within `radius` inside its own batch
    (pointgroup_ops.py BallQueryBatchP; bfs_cluster.cu:15-91).  The reference hands out CSR start positions with an
    atomicAdd (order = thread arrival); here they are the prefix sum in point order, one of its valid outcomes, with the
    same 1000-neighbour and n*meanActive caps."""
    require_cuda(coords, batch_idxs, batch_offsets)
    coords = coords.contiguous().float(); batch_idxs = batch_idxs.contiguous().int(); batch_offsets = batch_offsets.contiguous().int()
    n = coords.shape[0]
    counts = torch.zeros((n,), dtype=torch.int32, device=coords.device)
    lib = L.lib()
    check(lib.cg_pg_ballquery_batch_p(_p(coords), _p(batch_idxs), _p(batch_offsets), _c_int(n), ctypes.c_float(radius), _c_int(0), None, None,
                                      _p(counts), None, _stream()), 'cg_pg_ballquery_batch_p')
    start = (torch.cumsum(counts, 0) - counts).int()
    thre = n * meanActive
    length = torch.clamp(torch.minimum(counts, thre - start), min=0).int()          # writes are cut at n*meanActive (bfs_cluster.cu:52-57)
    total = int(min(int(counts.sum().item()), thre))
    idx = torch.zeros((max(thre, 1),), dtype=torch.int32, device=coords.device)
    check(lib.cg_pg_ballquery_batch_p(_p(coords), _p(batch_idxs), _p(batch_offsets), _c_int(n), ctypes.c_float(radius), _c_int(1), _p(start.contiguous()),
                                      _p(length.contiguous()), None, _p(idx), _stream()), 'cg_pg_ballquery_batch_p')
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
