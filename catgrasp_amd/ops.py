"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only)."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check, f32c, i32c, require_cuda

_c_int = ctypes.c_int
_c_long = ctypes.c_long

# bench.py hook: when set to {'mid_mode': m, 'events': []}, every cg_pointmlp_max launch with that mid_mode is
# bracketed by HIP events on the launch stream (torch's current stream) so the kernel's average duration can be
# measured live over the timed region; with a 'bgi_events' list, every cg_build_grasp_input launch too.
KERNEL_TIMER = None


def pointmlp_max(x, w1, b1, w2p, b2, w3p, b3, relu3, t3=None, mid_mode=0, wm=None, bm=None, t64=None,
                 nsplit=1, pointfeat=False, split=False, tile_points=256, status=None):
    """x:(B,N,6) -> (B,1024) [, pointfeat (B,N,64)].  See cg_pointmlp_max / cg_pointmlp_max_bf16x3 in
    include/catgrasp_amd.h.  split='f16' / 'bf16': w2p/w3p/wm are the split images of that element type (folding.pack_b_split);
    split='f16fp8': as 'f16' except that w3p is the f16fp8x2 image of the 128 -> 1024 layer (folding.pack_b_f16fp8x2)."""
    require_cuda(x)
    f32c(x)
    B, N, D = x.shape
    assert D == 6
    out = torch.empty((B, 1024), dtype=torch.float32, device=x.device)
    pf = torch.empty((B, N, 64), dtype=torch.float32, device=x.device) if pointfeat else None
    if split:
        timer = KERNEL_TIMER if (KERNEL_TIMER is not None and KERNEL_TIMER['mid_mode'] == mid_mode) else None
        if timer is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        args = (_p(x), _c_int(B), _c_int(N), _p(t3), _p(w1), _p(b1), _c_int(mid_mode), _p(wm), _p(bm),
                _p(t64), _p(w2p), _p(b2), _p(w3p), _p(b3), _c_int(int(relu3)), _c_int(nsplit),
                _c_int(tile_points), _p(out), _p(pf))
        if split == 'f16':      # status: caller-owned device int that collects the half range bits (None: not tracked)
            st = L.lib().cg_pointmlp_max_f16x3(*args, _p(status), _stream())
        elif split == 'f16fp8':
            st = L.lib().cg_pointmlp_max_f16fp8x2(*args, _p(status), _stream())
        else:
            st = L.lib().cg_pointmlp_max_bf16x3(*args, _stream())
        if timer is not None:
            ev1.record()
            timer['events'].append((ev0, ev1, (B, N)))
        check(st, {'f16': 'cg_pointmlp_max_f16x3', 'f16fp8': 'cg_pointmlp_max_f16fp8x2'}.get(split, 'cg_pointmlp_max_bf16x3'))
        return (out, pf) if pointfeat else out
    timer = KERNEL_TIMER if (KERNEL_TIMER is not None and KERNEL_TIMER['mid_mode'] == mid_mode) else None
    if timer is not None:
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    st = L.lib().cg_pointmlp_max(_p(x), _c_int(B), _c_int(N), _p(t3), _p(w1), _p(b1), _c_int(mid_mode), _p(wm), _p(bm),
                                 _p(t64), _p(w2p), _p(b2), _p(w3p), _p(b3), _c_int(int(relu3)), _c_int(nsplit),
                                 _p(out), _p(pf), _stream())
    if timer is not None:
        ev1.record()
        timer['events'].append((ev0, ev1, (B, N)))
    check(st, 'cg_pointmlp_max')
    return (out, pf) if pointfeat else out


def gemm_bias_act(x, wp, n_out, bias=None, relu=False, eye_k=0, row_bias=None, rows_per_group=1, split=False, status=None):
    """act(x @ W^T + bias [+ row_bias[row // rows_per_group]]) with packed W.  x:(M,K).
    split='f16' / 'bf16': wp is the split-packed image of that element type and the product runs on the split MFMA kernel."""
    require_cuda(x)
    f32c(x)
    M, K = x.shape
    y = torch.empty((M, n_out), dtype=torch.float32, device=x.device)
    ld_rb = row_bias.shape[1] if row_bias is not None else 0
    name = {'f16': 'cg_gemm_bias_act_f16x3', 'bf16': 'cg_gemm_bias_act_bf16x3'}.get(split, 'cg_gemm_bias_act')
    fn = getattr(L.lib(), name)
    args = (_p(x), _c_int(M), _c_int(K), _c_int(K), _p(wp), _c_int(n_out), _p(bias),
            _p(row_bias), _c_int(rows_per_group), _c_int(ld_rb), _c_int(int(relu)),
            _c_int(eye_k), _p(y), _c_int(n_out))
    st = fn(*args, _p(status), _stream()) if split == 'f16' else fn(*args, _stream())
    check(st, name)
    return y


def group_max(x, groups):
    """x (groups * rows, C) -> (groups, C): max over the rows of every group (cg_group_max)."""
    require_cuda(x)
    f32c(x)
    rows = x.shape[0] // max(groups, 1)
    assert groups * rows == x.shape[0]
    out = torch.empty((groups, x.shape[1]), dtype=torch.float32, device=x.device)
    check(L.lib().cg_group_max(_p(x), _c_long(groups), _c_long(rows), _c_int(x.shape[1]), _p(out), _stream()), 'cg_group_max')
    return out


def pose_inverse_rows_f64(poses, center, bad=None):
    """poses (E,16) float64 cuda (row-major 4x4, the caller's own numbers) -> (E,12) float32 rows of inv(pose) re-expressed for a cloud
    shifted by -center (cg_pose_inverse_rows_f64: float64 arithmetic, rounded once).  bad: optional (1,) int32 cuda flag collecting
    bits 1 (NaN / Inf in a pose), 2 (singular pose / inverse beyond float32), 4 (last row not 0 0 0 1); raise_bad_poses() turns it into
    the exception the reference's np.linalg.inv path would have ended in."""
    require_cuda(poses)
    assert poses.dtype == torch.float64 and poses.is_contiguous() and poses.shape[1] == 16
    out = torch.empty((poses.shape[0], 12), dtype=torch.float32, device=poses.device)
    c = (ctypes.c_double * 3)(*[float(v) for v in center])
    check(L.lib().cg_pose_inverse_rows_f64(_p(poses), _c_long(poses.shape[0]), c, _p(out), _p(bad), _stream()), 'cg_pose_inverse_rows_f64')
    return out


def raise_bad_poses(bits):
    """The exception for a non-zero flag word of pose_inverse_rows_f64 (dataset_grasp.py:69-70: np.linalg.inv(grasp_pose))."""
    bits = int(bits)
    if bits & 1:
        raise ValueError('grasp_poses contain NaN or Inf')
    if bits & 2:
        raise np.linalg.LinAlgError('Singular matrix')          # what np.linalg.inv raises in the reference's transform
    if bits & 4:
        raise ValueError('grasp_poses must be affine 4x4 transforms (last row 0 0 0 1)')


def softmax_pg(logits):
    """logits:(B,C) -> probs (B,C), label (B) int32, confidence (B), p_G (B)."""
    require_cuda(logits)
    f32c(logits)
    B, C = logits.shape
    probs = torch.empty_like(logits)
    label = torch.empty((B,), dtype=torch.int32, device=logits.device)
    conf = torch.empty((B,), dtype=torch.float32, device=logits.device)
    pg = torch.empty((B,), dtype=torch.float32, device=logits.device)
    check(L.lib().cg_softmax_pg(_p(logits), _c_int(B), _c_int(C), _p(probs), _p(label), _p(conf), _p(pg), _stream()),
          'cg_softmax_pg')
    return probs, label, conf, pg


def apply_shuffle_rows(partners, n_valid, n_pts, base=0, out=None):
    """partners: (count, stride) uint16 cuda -- the Fisher-Yates swap partners of `count` numpy permutation(n_valid) draws
    (transforms.NumpyChoiceStream.draw_partners) -> (count, n_pts) int32 = base + permutation[:n_pts] of every row."""
    require_cuda(partners)
    if partners.dtype != torch.uint16 or partners.dim() != 2 or not partners.is_contiguous():
        raise TypeError('partners must be a contiguous (count, stride) uint16 tensor')
    count, stride = partners.shape
    if out is None:
        out = torch.empty((count, n_pts), dtype=torch.int32, device=partners.device)
    check(L.lib().cg_apply_shuffle_rows(_p(partners), _c_long(stride), _c_int(n_valid), _c_int(n_pts), _c_long(count), _c_int(base), _p(out),
                                        _stream()), 'cg_apply_shuffle_rows')
    return out


def nunocs_decode(logits, nbins):
    """logits:(P,3*nbins) -> coords (P,3), conf_z (P)."""
    require_cuda(logits)
    f32c(logits)
    P = logits.shape[0]
    assert logits.shape[1] == 3 * nbins
    coords = torch.empty((P, 3), dtype=torch.float32, device=logits.device)
    conf = torch.empty((P,), dtype=torch.float32, device=logits.device)
    check(L.lib().cg_nunocs_decode(_p(logits), _c_long(P), _c_int(nbins), _p(coords), _p(conf), _stream()),
          'cg_nunocs_decode')
    return coords, conf


def build_grasp_input(cloud_xyz, cloud_normal, ids, pose_inv, mean=None, inv_std=None, out=None):
    """(n_cloud,3),(n_cloud,3),(G,n_pts) i32,(G,12) -> (G,n_pts,6)."""
    require_cuda(cloud_xyz, cloud_normal, ids, pose_inv)
    f32c(cloud_xyz); f32c(cloud_normal); i32c(ids); f32c(pose_inv)
    G, n_pts = ids.shape
    assert pose_inv.shape == (G, 12)
    if out is None:
        out = torch.empty((G, n_pts, 6), dtype=torch.float32, device=ids.device)
    timer = KERNEL_TIMER if (KERNEL_TIMER is not None and 'bgi_events' in KERNEL_TIMER) else None
    if timer is not None:
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    st = L.lib().cg_build_grasp_input(_p(cloud_xyz), _p(cloud_normal), _c_int(cloud_xyz.shape[0]), _p(ids), _p(pose_inv),
                                      _p(mean), _p(inv_std), _c_int(G), _c_int(n_pts), _p(out), _stream())
    if timer is not None:
        ev1.record()
        timer['bgi_events'].append((ev0, ev1, G))
    check(st, 'cg_build_grasp_input')
    return out


def build_nunocs_input(cloud_xyz, cloud_normal, ids, mean=None, inv_std=None):
    """(n_cloud,3),(n_cloud,3),(B,n_pts) i32 -> (B,n_pts,6)."""
    require_cuda(cloud_xyz, cloud_normal, ids)
    f32c(cloud_xyz); f32c(cloud_normal); i32c(ids)
    B, n_pts = ids.shape
    out = torch.empty((B, n_pts, 6), dtype=torch.float32, device=ids.device)
    check(L.lib().cg_build_nunocs_input(_p(cloud_xyz), _p(cloud_normal), _c_int(cloud_xyz.shape[0]), _p(ids),
                                        _p(mean), _p(inv_std), _c_int(B), _c_int(n_pts), _p(out), _stream()),
          'cg_build_nunocs_input')
    return out
