"""PointNet++ primitives of the reference's pointnet2.py:14-149 with the same tensor signatures,
executed by the HIP kernels in csrc/primitives.hip.  CUDA tensors only (no CPU fallback)."""
import ctypes

import torch

from . import _lib as L
from ._lib import _p, _stream, check, require_cuda

_c_int = ctypes.c_int
_c_long = ctypes.c_long


def _f32(t):
    return t.contiguous().float()


def square_distance(src, dst):
    """src (B,N,C), dst (B,M,C) -> (B,N,M) squared distances (-2ab + a^2 + b^2, pointnet2.py:30-32).  Any C, like the reference
    (its callers pass xyz: C = 3 takes the tiled kernel, other widths the generic one)."""
    require_cuda(src, dst)
    src = _f32(src); dst = _f32(dst)
    B, N, C = src.shape
    M = dst.shape[1]
    if dst.shape[0] != B or dst.shape[2] != C or C < 1:
        raise ValueError(f'square_distance: src {tuple(src.shape)} and dst {tuple(dst.shape)} do not match')
    out = torch.empty((B, N, M), dtype=torch.float32, device=src.device)
    check(L.lib().cg_square_distance_nd(_p(src), _p(dst), _c_int(B), _c_int(N), _c_int(M), _c_int(C), _p(out), _stream()), 'cg_square_distance_nd')
    return out


def _raise_if(err, what):
    if int(err.item()) != 0:
        raise IndexError(f'{what}: index out of range')


def index_points(points, idx):
    """points (B,N,C), idx (B,S) or (B,S,K) int64 -> (B,S[,K],C) (pointnet2.py:35-51)."""
    require_cuda(points, idx)
    points = _f32(points)
    idx = idx.contiguous().long()
    B, N, C = points.shape
    S = idx[0].numel() if B > 0 else 0
    out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=points.device)
    err = torch.zeros((1,), dtype=torch.int32, device=points.device)
    check(L.lib().cg_index_points(_p(points), _p(idx), _c_int(B), _c_int(N), _c_int(C), _c_long(S), _p(out), _p(err), _stream()),
          'cg_index_points')
    _raise_if(err, 'index_points')
    return out


def prepare_start(start, B, N, device):
    """The FPS start indices of one call, validated on the host and uploaded: what farthest_point_sample does at its top, available
    separately so that a stack can upload the starts of ALL its levels before it queues the first kernel (a pageable host-to-device
    copy in the middle of the stack waits for the stream to drain: the levels behind it would be queued one launch latency at a time).
    start None = the reference's draw, torch.randint(0, N, (B,)) on the CPU generator (pointnet2.py:66)."""
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long)
    start = torch.as_tensor(start).long()
    if start.numel() != B or (B > 0 and bool(((start < 0) | (start >= N)).any())):      # one read-back when `start` lives on the device, none otherwise
        raise ValueError('start must hold one valid point index per cloud')
    return start.to(device).contiguous()


def farthest_point_sample(xyz, npoint, start=None, return_xyz=False, start_prepared=False):
    """xyz (B,N,3) -> centroids (B,npoint) int64 (pointnet2.py:54-75).  `start` defaults to the reference's
    draw, `torch.randint(0, N, (B,), dtype=torch.long)` on the CPU generator (:66), so seeding torch
    reproduces the reference's samples exactly.  return_xyz=True also returns the sampled points,
    (B,npoint,3) == index_points(xyz, centroids), written by the same launch.  start_prepared: `start` is prepare_start's result."""
    require_cuda(xyz)
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    if C != 3:
        raise NotImplementedError('farthest_point_sample HIP kernel is built for 3-D points')
    if not start_prepared:           # start_prepared: the caller already ran prepare_start (validated device tensor)
        start = prepare_start(start, B, N, xyz.device)
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    scratch = torch.empty((B, N), dtype=torch.float32, device=xyz.device) if N > 24576 else None
    if return_xyz:
        new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device)
        check(L.lib().cg_farthest_point_sample_xyz(_p(xyz), _p(start), _c_int(B), _c_int(N), _c_int(npoint), _p(scratch), _p(out), _p(new_xyz),
                                                   _stream()), 'cg_farthest_point_sample_xyz')
        return out, new_xyz
    check(L.lib().cg_farthest_point_sample(_p(xyz), _p(start), _c_int(B), _c_int(N), _c_int(npoint), _p(scratch), _p(out), _stream()),
          'cg_farthest_point_sample')
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> group_idx (B,S,nsample) int64: first nsample in-radius indices in ascending order, padded with
    the first (pointnet2.py:78-98).  The reference slices a length-N sorted row, so nsample > N gives N
    columns; that degenerate case is reproduced."""
    require_cuda(xyz, new_xyz)
    xyz = _f32(xyz); new_xyz = _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    ns = min(int(nsample), N)
    out = torch.empty((B, S, ns), dtype=torch.int64, device=xyz.device)
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))     # the comparison is made in float32
    check(L.lib().cg_query_ball_point(_p(xyz), _p(new_xyz), _c_int(B), _c_int(N), _c_int(S), ctypes.c_float(r2), _c_int(ns), _p(out),
                                      _stream()), 'cg_query_ball_point')
    return out


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, start=None):
    """pointnet2.py:101-129: FPS -> gather centroids -> ball query -> grouped, centred xyz ++ point features."""
    require_cuda(xyz)
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    S = npoint
    fps_idx, new_xyz = farthest_point_sample(xyz, npoint, start, return_xyz=True)      # new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    K = idx.shape[2]
    D = 0
    if points is not None:
        points = _f32(points)
        D = points.shape[2]
    new_points = torch.empty((B, S, K, 3 + D), dtype=torch.float32, device=xyz.device)
    grouped_xyz = torch.empty((B, S, K, 3), dtype=torch.float32, device=xyz.device) if returnfps else None
    err = torch.zeros((1,), dtype=torch.int32, device=xyz.device)
    check(L.lib().cg_group_points(_p(xyz), _p(points), _p(new_xyz), _p(idx), _c_int(B), _c_int(N), _c_int(S), _c_int(K), _c_int(D),
                                  _p(new_points), _p(grouped_xyz), _p(err), _stream()), 'cg_group_points')
    _raise_if(err, 'sample_and_group (a query ball was empty)')
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """pointnet2.py:132-149: one group holding every point (a view + concat; no arithmetic)."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device, dtype=xyz.dtype)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        return new_xyz, torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped_xyz


class SetAbstractionWeights:
    """Folded (eval BatchNorm) + MFMA-packed weights of a shared per-neighbour MLP [Conv2d(1x1) -> BatchNorm2d -> ReLU] x L.
    kind 'reg'  : the register-resident / strip kernels of csrc/setabstraction.hip (3 + D <= 16 inputs, widths <= 256: the FIRST layer
                  of a PointNet++ stack); layer 0's columns are [xyz | features | zero pad to 16].
    kind 'tile' : csrc/sa_tile.hip (the layers past the first: wide inputs, hidden widths <= 512, last width any multiple of 32) and the
                  group-all layer's GEMM chain; layer 0's columns are [features | xyz | zero pad to a multiple of 8]."""

    TILE_MAX_CIN = 592          # 64 rows x (cin + 4) floats of LDS strip

    def __init__(self, layers, in_channel, device, kind=None):
        """layers: [(conv weight (Cout,Cin[,1,1]), conv bias (Cout), (bn weight, bn bias, running_mean, running_var) or None), ...]"""
        import numpy as np
        from . import folding
        if not 1 <= len(layers) <= 4:
            raise ValueError('1..4 layers')
        if in_channel < 3:
            raise ValueError('in_channel counts the 3 coordinates: >= 3')
        widths = [int(np.shape(w)[0]) for w, _, _ in layers]
        if any(c % 32 or c <= 0 for c in widths):
            raise NotImplementedError('fused set abstraction: layer widths must be multiples of 32')
        if kind is None:
            # first-level shapes: the register-resident kernel where it applies (<= 3 layers of 32 / 64 / 128 channels: 1.3 x the tile
            # kernel's rate there); its LDS-strip fallback only for narrow odd nets (<= 64 wide, e.g. four layers); everything else -- a
            # 256-wide layer, a width of 96, 3 + D > 16 -- the tile kernel (1.3 - 1.9 x the strip kernel: profiles/r5_sa_tile_vs_first_level_kernels.json)
            fits_reg = len(widths) <= 3 and all(c in (32, 64, 128) for c in widths)
            kind = 'reg' if in_channel <= 16 and (fits_reg or max(widths) <= 64) else 'tile'
        if kind == 'reg' and (in_channel > 16 or max(widths) > 256):
            raise ValueError("kind 'reg' holds 3 + D <= 16 inputs and widths <= 256")
        self.kind = kind
        self.cin, self.cout, self.w, self.b = [], [], [], []
        prev = in_channel
        for li, (w, b, bn) in enumerate(layers):
            w = np.asarray(w, dtype=np.float64).reshape(np.shape(w)[0], -1)
            if w.shape[1] != prev:
                raise ValueError(f'layer {li}: expected {prev} input channels, got {w.shape[1]}')
            wf, bf = folding.fold_bn(w, b, bn)
            if li == 0:
                if kind == 'reg':
                    wf = np.concatenate([wf, np.zeros((wf.shape[0], 16 - prev))], axis=1)
                else:
                    pad = (-prev) % 8
                    wf = np.concatenate([wf[:, 3:], wf[:, :3], np.zeros((wf.shape[0], pad))], axis=1)
            self.cin.append(wf.shape[1]); self.cout.append(wf.shape[0])
            self.w.append(torch.from_numpy(folding.pack_b(wf)).to(device))
            self.b.append(torch.from_numpy(bf.astype(np.float32)).to(device))
            prev = w.shape[0]
        self.in_channel = in_channel
        self.hidden_max = max(widths[:-1]) if len(widths) > 1 else 0

    def _c_arrays(self):
        L_ = len(self.w)
        return (L_, (ctypes.c_int * L_)(*self.cin), (ctypes.c_int * L_)(*self.cout),
                (ctypes.c_void_p * L_)(*[t.data_ptr() for t in self.w]), (ctypes.c_void_p * L_)(*[t.data_ptr() for t in self.b]))


def group_mlp_max(xyz, points, new_xyz, idx, W, check_indices=True, channels_last=False, out=None, append_xyz=0, err=None):
    """The consumer of sample_and_group fused into one kernel (cg_sa_group_mlp_max_strided / cg_sa_tile_mlp_max by W.kind): neighbourhoods
    idx (B,S,K) of xyz/points around new_xyz -> centred coordinates ++ features -> shared MLP W (SetAbstractionWeights) -> max over the K
    neighbours.  -> (B, C_out, S) float32, the layout torch.max(new_points, 2)[0] has in a PointNet++ set-abstraction layer, or with
    channels_last=True (B, S, C_out): the rows the next layer gathers from.  out: optional view to write into (any strides -- one scale's
    channel slice of a multi-scale layer's output).
    append_xyz = n (tile kernel, channels_last, `out` a view with >= n further channels behind it in memory): channels [C, C + n) of
    every row also receive new_xyz ++ zeros: the [features | xyz | pad] rows group_all_mlp_max(rows=...) reads.
    check_indices=False skips the read-back of the index-error flag (one host synchronisation per call) and returns (out, err_flag tensor):
    for callers that batch the check, and for timing the kernel alone."""
    require_cuda(xyz, new_xyz, idx)
    xyz = _f32(xyz); new_xyz = _f32(new_xyz)
    idx = idx.contiguous().long()
    B, N, _ = xyz.shape
    S, K = idx.shape[1], idx.shape[2]
    D = 0
    if points is not None:
        points = _f32(points); D = points.shape[2]
    if 3 + D != W.in_channel:
        raise ValueError(f'weights expect {W.in_channel} input channels, got 3 + {D}')
    C = W.cout[-1]
    if out is None:
        out = torch.empty((B, S, C) if channels_last else (B, C, S), dtype=torch.float32, device=xyz.device)
    elif tuple(out.shape) != ((B, S, C) if channels_last else (B, C, S)) or out.dtype != torch.float32 or not out.is_cuda:
        raise ValueError('out: wrong shape / dtype / device')
    sb, s1, s2 = out.stride() if B * S else (0, 0, 0)
    ss, cs = (s1, s2) if channels_last else (s2, s1)
    if err is None:            # err: a caller-owned, pre-zeroed (1,) int32 flag shared by several calls (a whole stack reads it back once)
        err = torch.zeros((1,), dtype=torch.int32, device=xyz.device)
    L_, cin, cout, wp, bp = W._c_arrays()
    if append_xyz and (W.kind != 'tile' or not channels_last or cs != 1):
        raise ValueError("append_xyz needs kind='tile' weights and a channels_last output with unit channel stride")
    if W.kind == 'reg':
        check(L.lib().cg_sa_group_mlp_max_strided(_p(xyz), _p(points), _p(new_xyz), _p(idx), _c_int(B), _c_int(N), _c_int(S), _c_int(K), _c_int(D),
                                                  _c_int(L_), cin, cout, wp, bp, _p(out), _c_long(sb), _c_long(ss), _c_long(cs), _p(err),
                                                  _stream()), 'cg_sa_group_mlp_max_strided')
    else:
        if W.cin[0] > W.TILE_MAX_CIN or W.hidden_max > 512:
            raise NotImplementedError('fused set abstraction (tile kernel): 3 + D <= 592 inputs, hidden widths <= 512')
        check(L.lib().cg_sa_tile_mlp_max(_p(xyz), _p(points), _p(new_xyz), _p(idx), _c_int(B), _c_int(N), _c_int(S), _c_int(K), _c_int(D),
                                         _c_int(L_), cin, cout, wp, bp, _p(out), _c_long(sb), _c_long(ss), _c_long(cs), _c_int(int(append_xyz)), _p(err),
                                         _stream()), 'cg_sa_tile_mlp_max')
    if not check_indices:
        return out, err
    _raise_if(err, 'group_mlp_max (a query ball was empty or an index is out of range)')
    return out


def group_all_mlp_max(xyz, points, W, fused=False, rows=None):
    """The group-all layer (sample_and_group_all, pointnet2.py:132-149, + the shared MLP + max over ALL points): xyz (B,N,3), points
    (B,N,D) | None -> (B, C_out).  One group per cloud means B * ceil(N / 64) row tiles: a handful for a PointNet++ head (N = 128), so
    the layers run as row-batched GEMMs over all B * N rows ([cg_sa_concat_input ->] cg_gemm_bias_act per hidden layer ->
    cg_gemm_bias_relu_groupmax: the 32 x 32 output tiles of every layer spread over the chip, the last layer's max over the points in
    its epilogue; the hidden activations, B * N x <= 512 floats, are the only intermediates).  rows: the (B * N, cin) input matrix
    [features | xyz | pad] when the previous level already wrote it (group_mlp_max(append_xyz=...)): no concatenation pass.
    fused=True: the fused tile kernel takes the layer whole (cg_sa_tile_mlp_max, idx == NULL) -- measured 6..9 x slower at 1..16 clouds
    of 128 points (profiles/r5_pp_encoder_first.json: two 64-row tiles per cloud carry 0.72 MMAC per row on one CU each), kept for the
    parity tests of the kernel's index-free mode."""
    from . import ops
    require_cuda(xyz)
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    D = 0
    if points is not None:
        D = points.shape[2]
        if rows is None or fused:
            points = _f32(points)
    if W.kind != 'tile' or 3 + D != W.in_channel:
        raise ValueError("group_all_mlp_max needs kind='tile' weights for 3 + D input channels")
    C = W.cout[-1]
    can_fuse = W.cin[0] <= W.TILE_MAX_CIN and W.hidden_max <= 512
    if fused:
        if not can_fuse:
            raise NotImplementedError('fused group-all layer: 3 + D <= 592 inputs, hidden widths <= 512')
        out = torch.empty((B, C), dtype=torch.float32, device=xyz.device)
        L_, cin, cout, wp, bp = W._c_arrays()
        check(L.lib().cg_sa_tile_mlp_max(_p(xyz), _p(points), _p(None), _p(None), _c_int(B), _c_int(N), _c_int(1), _c_int(N), _c_int(D),
                                         _c_int(L_), cin, cout, wp, bp, _p(out), _c_long(C), _c_long(0), _c_long(1), _c_int(0), _p(None), _stream()),
              'cg_sa_tile_mlp_max')
        return out
    if rows is not None:       # the previous level already wrote the [features | xyz | pad] rows (group_mlp_max(append_xyz=...))
        if tuple(rows.shape) != (B * N, W.cin[0]) or not rows.is_contiguous() or rows.dtype != torch.float32:
            raise ValueError(f'rows must be a contiguous float32 ({B * N}, {W.cin[0]}) tensor')
        h = rows
    else:
        h = torch.empty((B * N, W.cin[0]), dtype=torch.float32, device=xyz.device)
        check(L.lib().cg_sa_concat_input(_p(xyz), _p(points), _c_long(B * N), _c_int(D), _c_int(W.cin[0]), _p(h), _stream()), 'cg_sa_concat_input')
    for wp, b, co in zip(W.w[:-1], W.b[:-1], W.cout[:-1]):
        h = ops.gemm_bias_act(h, wp, co, bias=b, relu=True)
    out = torch.empty((B, C), dtype=torch.float32, device=xyz.device)       # last layer: the max over the N points in the GEMM's epilogue
    check(L.lib().cg_gemm_bias_relu_groupmax(_p(h), _c_int(B * N), _c_int(h.shape[1]), _c_int(h.shape[1]), _p(W.w[-1]), _c_int(C), _p(W.b[-1]),
                                             _c_int(N), _p(out), _stream()), 'cg_gemm_bias_relu_groupmax')
    return out
