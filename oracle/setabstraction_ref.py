"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the PointNet++ set-abstraction layers and the 3-level encoder the
package assembles from the reference's primitives.

The reference defines the primitives (pointnet2.py:54-149: farthest_point_sample, query_ball_point, sample_and_group,
sample_and_group_all) but never stacks them; BASELINE.json's north_star names the stack.  So the oracle of a
layer is: the restated primitives of oracle/pointnet_ref.py (pinned to the imported reference by tests/golden/pointnet2_golden.npz)
followed by plain torch float32 ops -- Conv2d(1x1) as a matmul, BatchNorm2d in eval form, ReLU, max over the neighbours -- on the
grouped tensor, exactly the op sequence the fused kernels replace.  Nothing here is imported by the product.

`idx` may be supplied per level: ball-query membership can flip inside the float rounding band of d^2 around r^2
(tests/test_primitives_gpu.py::test_query_ball_point), so the layer tests group the oracle on the device's neighbour lists after
checking that they differ from the oracle's own in < 1e-3 of the entries."""
import torch

from . import pointnet_ref as pr

BN_EPS = 1e-5


def mlp_max(new_points, layers, dtype=torch.float32):
    """new_points (B,S,K,C) -> [conv 1x1 -> BN(eval) -> ReLU] per layer -> max over K -> (B,S,C_last).
    layers: [(w (Co,Ci), b, bn_weight, bn_bias, running_mean, running_var), ...] tensors."""
    h = new_points.to(dtype)
    for w, b, g, beta, mu, var in layers:
        w, b, g, beta, mu, var = [t.to(dtype) for t in (w, b, g, beta, mu, var)]
        h = h @ w.reshape(w.shape[0], -1).t() + b                     # Conv2d(1x1) over the channel axis
        h = (h - mu) / torch.sqrt(var + BN_EPS) * g + beta            # BatchNorm2d, eval
        h = torch.relu(h)
    return h.max(dim=2)[0]


def layers_of(sd, prefix, n, conv='mlp_convs', bn='mlp_bns'):
    return [(sd[f'{prefix}{conv}.{i}.weight'], sd[f'{prefix}{conv}.{i}.bias'], sd[f'{prefix}{bn}.{i}.weight'], sd[f'{prefix}{bn}.{i}.bias'],
             sd[f'{prefix}{bn}.{i}.running_mean'], sd[f'{prefix}{bn}.{i}.running_var']) for i in range(n)]


def sa_forward(xyz, points, npoint, radius, nsample, layers, start, idx=None, dtype=torch.float32):
    """One single-scale level: sample_and_group (pointnet2.py:101-129) -> mlp_max.  -> new_xyz (B,S,3), new_points (B,S,C), fps_idx, idx"""
    B = xyz.shape[0]
    fps_idx = pr.farthest_point_sample(xyz, npoint, start)
    new_xyz = pr.index_points(xyz, fps_idx)
    if idx is None:
        idx = pr.query_ball_point(radius, nsample, xyz, new_xyz)
    grouped = pr.index_points(xyz, idx) - new_xyz.view(B, npoint, 1, 3)
    if points is not None:
        grouped = torch.cat([grouped, pr.index_points(points, idx)], dim=-1)
    return new_xyz, mlp_max(grouped, layers, dtype), fps_idx, idx


def sa_all_forward(xyz, points, layers, dtype=torch.float32):
    """The group-all level: sample_and_group_all (pointnet2.py:132-149) -> mlp_max.  -> (B, C)"""
    _, new_points = pr.sample_and_group_all(xyz, points)
    return mlp_max(new_points, layers, dtype)[:, 0]


def sa_msg_forward(xyz, points, npoint, radius_list, nsample_list, layers_list, start, idx_list=None, dtype=torch.float32):
    """One multi-scale level: one FPS, per scale ball query -> group -> mlp_max, concatenated along the channels."""
    B = xyz.shape[0]
    fps_idx = pr.farthest_point_sample(xyz, npoint, start)
    new_xyz = pr.index_points(xyz, fps_idx)
    outs, idxs = [], []
    for i, (radius, K, layers) in enumerate(zip(radius_list, nsample_list, layers_list)):
        idx = pr.query_ball_point(radius, K, xyz, new_xyz) if idx_list is None else idx_list[i]
        grouped = pr.index_points(xyz, idx) - new_xyz.view(B, npoint, 1, 3)
        if points is not None:
            grouped = torch.cat([grouped, pr.index_points(points, idx)], dim=-1)
        outs.append(mlp_max(grouped, layers, dtype)); idxs.append(idx)
    return new_xyz, torch.cat(outs, dim=-1), fps_idx, idxs
