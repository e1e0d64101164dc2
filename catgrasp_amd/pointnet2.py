"""Drop-in for the reference's `pointnet2` module (star-imported at predicter.py:19,
trainer_grasp.py:15, trainer_nunocs.py:16, run_grasp_simulation.py:26).

Same public names, constructor signatures, forward signatures and -- crucially -- the same
parameter / buffer names (``feat.stn.conv1.weight`` ... ``fc3.bias``), so reference checkpoints load
through ``Utils.load_model`` unchanged.

Execution:
  * eval mode + no grad + CUDA tensor  -> hand-written HIP kernels (catgrasp_amd.engine); weights are
    BN-folded / packed once per parameter version and cached on the device.
  * training mode (or grad enabled)    -> ordinary differentiable torch ops, so the reference trainers
    keep working (training is out of scope of the HIP path).
  * eval mode on a CPU tensor           -> RuntimeError: there is no CPU inference fallback.
The same rule holds for the building blocks used on their own: STN3d and PointNetEncoder (channel = 3..6, with or without the
feature transform -- the reference defaults are channel=3, feature_transform=False) dispatch to the fused HIP passes; a
free-standing STNkd(k) runs its layers on the HIP GEMM kernel (inside the encoder it is fused into pass <1>).  More than 6 input
channels raise NotImplementedError ("contract limits").
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine, folding
from . import primitives as _prim

__all__ = ['square_distance', 'index_points', 'farthest_point_sample', 'query_ball_point', 'sample_and_group',
           'sample_and_group_all', 'STN3d', 'STNkd', 'PointNetEncoder', 'PointNetCls', 'PointNetSeg', 'PointNetSetAbstraction', 'PointNetSetAbstractionMsg',
           'PointNet2Encoder']

# PointNet++ primitives (pointnet2.py:14-149): HIP implementations with the reference's tensor signatures
square_distance = _prim.square_distance
index_points = _prim.index_points
farthest_point_sample = _prim.farthest_point_sample
query_ball_point = _prim.query_ball_point
sample_and_group = _prim.sample_and_group
sample_and_group_all = _prim.sample_and_group_all


# The fused kernels max-pool with NaN-ignoring arithmetic (-fno-honor-nans): where the reference would hand back NaN outputs, a NaN / Inf
# input point would be pooled away silently -- so every eval-mode forward first rejects non-finite input (one reduction over the input +
# a 1-byte read-back, i.e. one host synchronisation per call).  A caller that guarantees finite inputs and wants its forwards to stay
# asynchronous on the stream can switch the check off: catgrasp_amd.pointnet2.VALIDATE_INPUTS = False  (or CATGRASP_AMD_VALIDATE_INPUTS=0).
# (The predicters do not pay for it: they validate the cloud once on the host, transforms.DeviceCloud.)
import os as _os
VALIDATE_INPUTS = _os.environ.get('CATGRASP_AMD_VALIDATE_INPUTS', '1') != '0'


def _use_hip(module, x, validated=False):
    """validated: the caller (a stack) has already checked this tensor's source for NaN / Inf -- the check is a host synchronisation, and a
    level's coordinates are copies of input coordinates that passed it."""
    if module.training or torch.is_grad_enabled():
        return False
    if not x.is_cuda:
        raise RuntimeError('catgrasp_amd.pointnet2: eval-mode inference needs a CUDA/HIP tensor '
                           '(the HIP kernels are the only inference path; there is no CPU fallback)')
    if VALIDATE_INPUTS and not validated and not bool(torch.isfinite(x).all()):
        raise ValueError('catgrasp_amd.pointnet2: the input contains NaN or Inf')
    return True


def _cached_weights(module, device, prepare):
    """Folded/packed device weights of `module`, rebuilt when a parameter/buffer changed (tensor version counters) or moved."""
    key = (str(device),) + tuple(int(t._version) for t in module.state_dict().values())
    cache = module.__dict__.get('_cg_cache')
    if cache is None or cache[0] != key:
        cache = (key, prepare(module.state_dict(), device))
        module.__dict__['_cg_cache'] = cache
    return cache[1]


class _TNet(nn.Module):
    """Shared body of STN3d / STNkd: per-point MLP cin->64->128->1024, max-pool, 1024->512->256->k*k, + I."""

    def __init__(self, cin, k):
        super().__init__()
        widths = [cin, 64, 128, 1024]
        for i in range(3):
            setattr(self, f'conv{i + 1}', nn.Conv1d(widths[i], widths[i + 1], 1))
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        for i, c in enumerate([64, 128, 1024, 512, 256]):
            setattr(self, f'bn{i + 1}', nn.BatchNorm1d(c))
        self._k = k

    def forward(self, x):
        if _use_hip(self, x):
            return self._hip_forward(x)
        return self._torch_forward(x)

    def _torch_forward(self, x):
        for i in (1, 2, 3):
            x = F.relu(getattr(self, f'bn{i}')(getattr(self, f'conv{i}')(x)))
        x = x.max(dim=2)[0]
        x = F.relu(self.bn4(self.fc1(x)))
        x = F.relu(self.bn5(self.fc2(x)))
        x = self.fc3(x) + torch.eye(self._k, device=x.device, dtype=x.dtype).reshape(1, -1)
        return x.view(-1, self._k, self._k)


class STN3d(_TNet):
    def __init__(self, channel):
        super().__init__(channel, 3)
        self._channel = channel

    def _hip_forward(self, x):
        """Eval-mode inference of a standalone STN3d: the fused STN pass + FC tail (engine.stn3d_forward)."""
        W = _cached_weights(self, x.device, folding.prepare_stn3d)       # channel = 3..6 (more: NotImplementedError from the fold)
        xt = engine.pad_points(x.float().transpose(1, 2).contiguous(), W.cin)
        return engine.run_guarded_features(engine.stn3d_forward, W, xt).view(-1, 3, 3)


class STNkd(_TNet):
    def __init__(self, k=64):
        super().__init__(k, k)
        self.k = k

    def _hip_forward(self, x):
        """Eval-mode inference of a free-standing STNkd on an arbitrary (B,k,N) tensor (pointnet2.py:206-223): the fused pass that
        contains STNkd inside the encoder starts from the 6-channel points, so here every layer runs on the HIP GEMM kernel, with the
        max over points on cg_group_max (engine.stnkd_forward; exact f32 in every arithmetic mode)."""
        W = _cached_weights(self, x.device, folding.prepare_stnkd)
        return engine.stnkd_forward(W, x.float().transpose(1, 2).contiguous()).view(-1, self.k, self.k)


class PointNetEncoder(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        self.stn = STN3d(channel)
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if feature_transform:
            self.fstn = STNkd(k=64)

    def forward(self, x):
        """x:(B,D,N).  Eval-mode inference on a HIP tensor runs the fused passes (the same kernels PointNetCls / PointNetSeg use);
        training / grad-enabled calls use the differentiable torch ops below."""
        if _use_hip(self, x):
            W = _cached_weights(self, x.device, lambda sd, dev: folding.prepare_encoder(sd, '', dev))
            xt = engine.pad_points(x.float().transpose(1, 2).contiguous(), W.cin)
            gf = self.global_feat
            return engine.run_guarded_features(lambda w, xx, st: engine.encoder_module_forward(w, xx, gf, st), W, xt)
        return self._torch_forward(x)

    def _torch_forward(self, x):
        B, D, N = x.shape
        trans = self.stn._torch_forward(x)
        pts = x.transpose(2, 1)
        xyz = torch.bmm(pts[:, :, :3], trans)
        pts = torch.cat([xyz, pts[:, :, 3:]], dim=2) if D > 3 else xyz
        h = F.relu(self.bn1(self.conv1(pts.transpose(2, 1))))
        trans_feat = None
        if self.feature_transform:
            trans_feat = self.fstn._torch_forward(h)
            h = torch.bmm(h.transpose(2, 1), trans_feat).transpose(2, 1)
        pointfeat = h
        h = F.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        g = h.max(dim=2)[0]
        if self.global_feat:
            return g, trans, trans_feat
        return torch.cat([g.unsqueeze(2).expand(-1, -1, N), pointfeat], 1), trans, trans_feat


class _HipCached(nn.Module):
    """Caches the folded/packed device weights, keyed on parameter versions + device."""

    _kind = None

    def _device_weights(self, device):
        return _cached_weights(self, device, folding.prepare_cls if self._kind == 'cls' else folding.prepare_seg)


class PointNetCls(_HipCached):
    """Grasp-Q classifier (pointnet2.py:275-299).  forward(x:(B,N,D)) -> (logits (B,n_out), trans_feat (B,64,64))."""
    _kind = 'cls'

    def __init__(self, n_in, n_out):
        super().__init__()
        self.feat = PointNetEncoder(global_feat=True, feature_transform=True, channel=n_in)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, n_out)
        self.dropout = nn.Dropout(p=0.4)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()
        self._n_in = n_in

    def forward(self, x):
        if _use_hip(self, x):
            W = self._device_weights(x.device)                               # n_in = 3..6 (config_grasp.yml: 6)
            return engine.run_guarded(engine.cls_forward, W, engine.pad_points(x.float().contiguous(), W.cin))
        g, _, trans_feat = self.feat._torch_forward(x.permute(0, 2, 1))
        h = F.relu(self.bn1(self.fc1(g)))
        h = F.relu(self.bn2(self.fc2(self.dropout(h))))
        return self.fc3(h), trans_feat


class PointNetSeg(_HipCached):
    """NUNOCS per-point classifier (pointnet2.py:302-329).  forward(x:(B,N,D)) -> ((B,N,n_out), trans_feat)."""
    _kind = 'seg'

    def __init__(self, n_in, n_out):
        super().__init__()
        self.feat = PointNetEncoder(global_feat=False, feature_transform=True, channel=n_in)
        self.conv1 = nn.Conv1d(1088, 512, 1)
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.conv3 = nn.Conv1d(256, 128, 1)
        self.conv4 = nn.Conv1d(128, n_out, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(128)
        self._n_in = n_in

    def forward(self, x):
        if _use_hip(self, x):
            W = self._device_weights(x.device)                               # n_in = 3..6 (config_nunocs.yml: 6)
            return engine.run_guarded(engine.seg_forward, W, engine.pad_points(x.float().contiguous(), W.cin))
        f, _, trans_feat = self.feat._torch_forward(x.permute(0, 2, 1))
        h = F.relu(self.bn1(self.conv1(f)))
        h = F.relu(self.bn2(self.conv2(h)))
        h = F.relu(self.bn3(self.conv3(h)))
        return self.conv4(h).permute(0, 2, 1), trans_feat


def _sa_layers_from_state(sd, prefix, n):
    g = lambda k: sd[k].detach().cpu().double().numpy()
    return [(g(f'{prefix}convs.{i}.weight'), g(f'{prefix}convs.{i}.bias'),
             tuple(g(f'{prefix}bns.{i}.{k}') for k in ('weight', 'bias', 'running_mean', 'running_var'))) for i in range(n)]


def _mlp_torch(h, convs, bns):
    for conv, bn in zip(convs, bns):
        h = F.relu(bn(conv(h)))
    return h


class PointNetSetAbstraction(nn.Module):
    """The set-abstraction layer the reference's primitives were written for (pointnet2.py:14-149 define sample_and_group and
    friends, but the reference never assembles them into a layer -- SURVEY.md §0 F1; BASELINE.json's north_star names it):
        new_xyz, new_points = sample_and_group(npoint, radius, nsample, xyz, points)      [group_all: sample_and_group_all(xyz, points)]
        new_points -> (B, 3+D, K, S) -> [Conv2d(1x1) -> BatchNorm2d -> ReLU] per mlp width -> max over the K neighbours
    forward(xyz (B,N,3), points (B,N,D) | None) -> (new_xyz (B,S,3), new_points (B,S,C_last)); in_channel = 3 + D.
    Eval-mode inference on a HIP tensor: FPS + ball query + ONE fused group->MLP->max kernel (primitives.group_mlp_max: the
    register-resident kernel for a first layer, the LDS-tile kernel for 3 + D > 16 inputs or wider layers), the grouped tensor never
    exists; group_all: primitives.group_all_mlp_max.  Training / grad-enabled calls use the torch ops on the grouped tensor."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all=False):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.in_channel, self.group_all = npoint, radius, nsample, in_channel, group_all
        self.mlp_convs = nn.ModuleList(); self.mlp_bns = nn.ModuleList()
        last = in_channel
        for c in mlp:
            self.mlp_convs.append(nn.Conv2d(last, c, 1)); self.mlp_bns.append(nn.BatchNorm2d(c))
            last = c

    def _weights(self, device):
        n = len(self.mlp_convs)
        kind = 'tile' if self.group_all else None
        return _cached_weights(self, device, lambda sd, dev: _prim.SetAbstractionWeights(_sa_layers_from_state(sd, 'mlp_', n), self.in_channel,
                                                                                          dev, kind=kind))

    def forward(self, xyz, points, start=None, _err=None, _rows=None, _new_xyz=None, _validated=False):
        """_new_xyz (stack-internal): the level's sampled points when the stack has already run its farthest-point sampling (on a side stream).
        _err: a pre-zeroed (1,) int32 device flag shared by the levels of a stack (one read-back for the stack instead of one per layer).
        _rows (stack-internal): for a sampling level, a (B, S, roundup8(C + 3)) buffer to produce the output in -- features in
        [..., :C] (the returned new_points is that view), the level's new_xyz ++ zeros behind them, i.e. the input rows of a following
        group-all level; for the group-all level, that buffer.
        _validated (stack-internal): the stack has checked its input for NaN / Inf once (no further host synchronisation per level)."""
        if _use_hip(self, xyz, _validated):
            W = self._weights(xyz.device)
            if self.group_all:
                B = xyz.shape[0]
                rows = None if _rows is None else _rows.view(-1, _rows.shape[-1])
                return (torch.zeros((B, 1, 3), dtype=torch.float32, device=xyz.device),
                        _prim.group_all_mlp_max(xyz, points, W, rows=rows).view(B, 1, -1))
            new_xyz = _new_xyz
            if new_xyz is None:
                _, new_xyz = farthest_point_sample(xyz, self.npoint, start, return_xyz=True)      # = index_points(xyz, fps_idx), same launch
            idx = query_ball_point(self.radius, self.nsample, xyz, new_xyz)
            kw = {}
            if _rows is not None:
                C = W.cout[-1]
                kw = {'out': _rows[:, :, :C]}
                if W.kind == 'tile':
                    kw['append_xyz'] = _rows.shape[-1] - C
                else:                      # a first-layer shape feeding a group-all level directly: the three columns by a copy
                    _rows[:, :, C:C + 3] = new_xyz; _rows[:, :, C + 3:] = 0
            if _err is not None:
                return new_xyz, _prim.group_mlp_max(xyz, points, new_xyz, idx, W, check_indices=False, channels_last=True, err=_err, **kw)[0]
            return new_xyz, _prim.group_mlp_max(xyz, points, new_xyz, idx, W, channels_last=True, **kw)
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
        else:       # indices from the HIP kernels (or torch ops on a CPU tensor); the gathers are differentiable torch indexing
            fps_idx = farthest_point_sample(xyz, self.npoint, start) if xyz.is_cuda else _torch_fps(xyz, self.npoint, start)
            new_xyz = _torch_index(xyz, fps_idx)
            idx = query_ball_point(self.radius, self.nsample, xyz, new_xyz) if xyz.is_cuda else _torch_ball(self.radius, self.nsample, xyz, new_xyz)
            new_points = _torch_index(xyz, idx) - new_xyz.unsqueeze(2)
            if points is not None:
                new_points = torch.cat([new_points, _torch_index(points, idx)], dim=-1)
        h = _mlp_torch(new_points.permute(0, 3, 2, 1), self.mlp_convs, self.mlp_bns)          # (B, 3+D, K, S) -> (B, C, K, S)
        return new_xyz, torch.max(h, 2)[0].permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """Multi-scale grouping: ONE farthest-point sample, then per scale i  query_ball_point(radius_list[i], nsample_list[i]) -> group ->
    shared MLP mlp_list[i] -> max, the scales' outputs concatenated along the channels (the multi-scale layer the primitives of
    pointnet2.py:54-129 build; north_star: "set-abstraction encoder").  in_channel = D, the feature channels WITHOUT the 3 coordinates
    (the convention of the usual PointNet++ code for this layer).  forward(xyz (B,N,3), points (B,N,D) | None) ->
    (new_xyz (B,S,3), new_points (B,S,sum_i mlp_list[i][-1])).  HIP path: every scale's fused kernel writes its channel slice of the
    output directly (strided store); nothing is concatenated."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        assert len(radius_list) == len(nsample_list) == len(mlp_list)
        self.npoint, self.radius_list, self.nsample_list, self.in_channel = npoint, list(radius_list), list(nsample_list), in_channel
        self.conv_blocks = nn.ModuleList(); self.bn_blocks = nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel + 3
            for c in mlp:
                convs.append(nn.Conv2d(last, c, 1)); bns.append(nn.BatchNorm2d(c))
                last = c
            self.conv_blocks.append(convs); self.bn_blocks.append(bns)
        self.out_channel = sum(m[-1] for m in mlp_list)

    def forward(self, xyz, points, start=None, _err=None, _rows=None, _new_xyz=None, _validated=False):
        if _use_hip(self, xyz, _validated):
            def prep(sd, dev):
                out = []
                for i, convs in enumerate(self.conv_blocks):
                    g = lambda k: sd[k].detach().cpu().double().numpy()
                    layers = [(g(f'conv_blocks.{i}.{j}.weight'), g(f'conv_blocks.{i}.{j}.bias'),
                               tuple(g(f'bn_blocks.{i}.{j}.{k}') for k in ('weight', 'bias', 'running_mean', 'running_var'))) for j in range(len(convs))]
                    out.append(_prim.SetAbstractionWeights(layers, self.in_channel + 3, dev))
                return out
            Ws = _cached_weights(self, xyz.device, prep)
            B = xyz.shape[0]
            new_xyz = _new_xyz
            if new_xyz is None:
                _, new_xyz = farthest_point_sample(xyz, self.npoint, start, return_xyz=True)
            buf = _rows if _rows is not None else torch.empty((B, self.npoint, self.out_channel), dtype=torch.float32, device=xyz.device)
            out = buf[:, :, :self.out_channel]
            err = torch.zeros((1,), dtype=torch.int32, device=xyz.device) if _err is None else _err
            c0 = 0
            for i, (W, radius, K) in enumerate(zip(Ws, self.radius_list, self.nsample_list)):
                idx = query_ball_point(radius, K, xyz, new_xyz)
                last = i == len(Ws) - 1 and _rows is not None and W.kind == 'tile'      # the last scale's kernel also writes xyz ++ pad
                _prim.group_mlp_max(xyz, points, new_xyz, idx, W, check_indices=False, channels_last=True, out=out[:, :, c0:c0 + W.cout[-1]],
                                    append_xyz=buf.shape[-1] - self.out_channel if last else 0, err=err)
                c0 += W.cout[-1]
            if _rows is not None and Ws[-1].kind != 'tile':
                buf[:, :, self.out_channel:self.out_channel + 3] = new_xyz; buf[:, :, self.out_channel + 3:] = 0
            if _err is None:
                _prim._raise_if(err, 'PointNetSetAbstractionMsg (a query ball was empty or an index is out of range)')
            return new_xyz, out
        fps_idx = farthest_point_sample(xyz, self.npoint, start) if xyz.is_cuda else _torch_fps(xyz, self.npoint, start)
        new_xyz = _torch_index(xyz, fps_idx)
        outs = []
        for convs, bns, radius, K in zip(self.conv_blocks, self.bn_blocks, self.radius_list, self.nsample_list):
            idx = query_ball_point(radius, K, xyz, new_xyz) if xyz.is_cuda else _torch_ball(radius, K, xyz, new_xyz)
            grouped = _torch_index(xyz, idx) - new_xyz.unsqueeze(2)
            if points is not None:
                grouped = torch.cat([grouped, _torch_index(points, idx)], dim=-1)
            outs.append(torch.max(_mlp_torch(grouped.permute(0, 3, 2, 1), convs, bns), 2)[0])
        return new_xyz, torch.cat(outs, dim=1).permute(0, 2, 1)


def _torch_index(points, idx):
    """index_points (pointnet2.py:35-51) in differentiable torch ops (training path of the multi-scale layer)."""
    B = points.shape[0]
    view = [B] + [1] * (idx.dim() - 1)
    b = torch.arange(B, device=points.device).view(view).expand_as(idx)
    return points[b, idx]


def _torch_fps(xyz, npoint, start):
    """farthest_point_sample (pointnet2.py:54-75) in torch ops: CPU tensors in training mode only."""
    B, N, _ = xyz.shape
    cent = torch.zeros(B, npoint, dtype=torch.long, device=xyz.device)
    dist = torch.full((B, N), 1e10, device=xyz.device)
    far = torch.randint(0, N, (B,), dtype=torch.long) if start is None else torch.as_tensor(start).long()
    far = far.to(xyz.device)
    bi = torch.arange(B, device=xyz.device)
    for i in range(npoint):
        cent[:, i] = far
        d = ((xyz - xyz[bi, far].view(B, 1, 3)) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = dist.max(-1)[1]
    return cent


def _torch_ball(radius, nsample, xyz, new_xyz):
    """query_ball_point (pointnet2.py:78-98) in torch ops: CPU tensors in training mode only."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = -2 * new_xyz @ xyz.transpose(1, 2) + (new_xyz ** 2).sum(-1, keepdim=True) + (xyz ** 2).sum(-1).unsqueeze(1)
    idx = torch.arange(N, device=xyz.device).view(1, 1, N).repeat(B, S, 1)
    idx[d > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, :1].expand_as(idx)
    return torch.where(idx == N, first, idx)


_SIDE_STREAMS = {}
SIDE_STREAM_MIN_CLOUDS = int(_os.environ.get('CATGRASP_AMD_ENCODER_SIDE_STREAM_MIN', 12))     # PointNet2Encoder: batches from this size on overlap level 2's sampling with level 1's grouping


def _side_stream(device):
    """One extra HIP stream per device for work a stack overlaps with its main stream (PointNet2Encoder: the next level's sampling)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class PointNet2Encoder(nn.Module):
    """The PointNet++ set-abstraction ENCODER BASELINE.json's north_star names, assembled from the reference's primitives
    (pointnet2.py:54-149): three set-abstraction levels ending in one group over everything that is left.
        single-scale (default):  SA(512, r 0.2, K 32, [64, 64, 128]) -> SA(128, r 0.4, K 64, [128, 128, 256]) -> SA(all, [256, 512, 1024])
        multi-scale (msg=True):  SA(512, r [0.1, 0.2, 0.4], K [16, 32, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
                              -> SA(128, r [0.2, 0.4, 0.8], K [32, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]]) -> SA(all, [256, 512, 1024])
    forward(x (B,N,channel) -- coordinates first, like PointNetCls.forward's x; channel = 3 + features)
        -> (global feature (B,1024), [(l1_xyz, l1_points), (l2_xyz, l2_points)])
    `start`: per-level FPS start indices [(B,), (B,)] (default: the reference's torch.randint draw per level, pointnet2.py:66).
    Eval mode on a HIP tensor: per level one FPS launch, one ball query and one fused group->MLP->max kernel per scale, the levels'
    outputs written as the (B,S,C) rows the next level gathers from; the index-error flags of all levels are read back once."""

    def __init__(self, channel=6, msg=False, npoints=(512, 128), radii=None, nsamples=None, mlps=None):
        super().__init__()
        D = channel - 3
        self.channel, self.msg = channel, msg
        if not msg:
            radii = radii or (0.2, 0.4); nsamples = nsamples or (32, 64)
            mlps = mlps or ((64, 64, 128), (128, 128, 256), (256, 512, 1024))
            self.sa1 = PointNetSetAbstraction(npoints[0], radii[0], nsamples[0], 3 + D, list(mlps[0]))
            self.sa2 = PointNetSetAbstraction(npoints[1], radii[1], nsamples[1], 3 + mlps[0][-1], list(mlps[1]))
            c2 = mlps[1][-1]
        else:
            radii = radii or ((0.1, 0.2, 0.4), (0.2, 0.4, 0.8)); nsamples = nsamples or ((16, 32, 128), (32, 64, 128))
            mlps = mlps or (((32, 32, 64), (64, 64, 128), (64, 96, 128)), ((64, 64, 128), (128, 128, 256), (128, 128, 256)), (256, 512, 1024))
            self.sa1 = PointNetSetAbstractionMsg(npoints[0], radii[0], nsamples[0], D, [list(m) for m in mlps[0]])
            self.sa2 = PointNetSetAbstractionMsg(npoints[1], radii[1], nsamples[1], self.sa1.out_channel, [list(m) for m in mlps[1]])
            c2 = self.sa2.out_channel
        self.sa3 = PointNetSetAbstraction(None, None, None, 3 + c2, list(mlps[2]), group_all=True)
        self.out_channel = mlps[2][-1]

    def forward(self, x, start=None):
        B, N, C = x.shape
        if C != self.channel:
            raise ValueError(f'expected {self.channel} input channels, got {C}')
        xyz = x[:, :, :3].contiguous()
        feats = x[:, :, 3:].contiguous() if C > 3 else None
        s1, s2 = (None, None) if start is None else start
        # the fused stack needs EVERY level in eval mode: a level left in train mode (partial fine-tuning) takes its torch branch, which
        # ignores the stack-internal arguments -- then each level decides for itself below
        hip = not any(m.training for m in (self.sa1, self.sa2, self.sa3)) and _use_hip(self, x)
        err = torch.zeros((1,), dtype=torch.int32, device=x.device) if hip else None
        kw = {'_err': err, '_validated': True} if hip else {}       # x was checked once, above: no stream drain between the levels
        rows = None
        if hip:
            # Level 2's sampling chain needs level 1's sampled POINTS only, not its features: from SIDE_STREAM_MIN_CLOUDS clouds on it runs on
            # a side stream while level 1's ball query and fused kernel use the rest of the chip (a sampling chain occupies one CU per
            # cloud).  Measured (profiles/r5_pp_encoder_side_stream.json): 16 clouds 0.973 -> 0.948 ms, 8 clouds unchanged, ONE cloud
            # 0.771 -> 0.830 ms -- the two cross-stream waits cost more than the 41 us of level-1 work they hide -- hence the threshold.
            # both levels' FPS starts are drawn / validated / uploaded BEFORE the first kernel is queued (the reference's draw order, level
            # 1 then level 2, is kept): a pageable upload between the levels would wait for the stream to drain
            s1 = _prim.prepare_start(s1, B, N, x.device)
            s2 = _prim.prepare_start(s2, B, self.sa1.npoint, x.device)
            fps = lambda pts, n, st: farthest_point_sample(pts, n, st, return_xyz=True, start_prepared=True)
            _, l1_xyz = fps(xyz, self.sa1.npoint, s1)
            if B >= SIDE_STREAM_MIN_CLOUDS:
                cur, side = torch.cuda.current_stream(x.device), _side_stream(x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _, l2_xyz = fps(l1_xyz, self.sa2.npoint, s2)
                _, l1_points = self.sa1(xyz, feats, _new_xyz=l1_xyz, **kw)
                cur.wait_stream(side)
                l2_xyz.record_stream(cur)
            else:
                _, l1_points = self.sa1(xyz, feats, _new_xyz=l1_xyz, **kw)
                _, l2_xyz = fps(l1_xyz, self.sa2.npoint, s2)
            # level 2 writes [features | xyz | pad] rows: what the group-all level's first GEMM reads (no concatenation pass)
            c2 = self.sa3.in_channel - 3
            rows = torch.empty((B, self.sa2.npoint, (c2 + 3 + 7) & ~7), dtype=torch.float32, device=x.device)
            _, l2_points = self.sa2(l1_xyz, l1_points, _new_xyz=l2_xyz, _rows=rows, **kw)
        else:
            l1_xyz, l1_points = self.sa1(xyz, feats, start=s1)
            l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, start=s2)
        _, l3_points = self.sa3(l2_xyz, l2_points, **({'_rows': rows, '_validated': True} if rows is not None else {}))
        if hip:
            _prim._raise_if(err, 'PointNet2Encoder (a query ball was empty or an index is out of range)')
        return l3_points.reshape(B, -1), [(l1_xyz, l1_points), (l2_xyz, l2_points)]
