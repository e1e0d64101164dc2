"""Drop-in for the reference's `aligning.estimate9DTransform` (aligning.py:83-119) with the hypothesis loop on
the device (csrc/ransac.hip): 10,000 sequential `cv2.estimateAffine3D` + numpy passes become one kernel launch."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int


def draw_hypothesis_ids(n_points, max_iter):
    """The reference's sampling (aligning.py:89-93): one `np.random.choice(n, size=4, replace=False)` per iteration from
    numpy's global generator, so a seeded run reproduces the reference's hypotheses.  -> (max_iter,4) int32."""
    from . import transforms
    if n_points < 4:
        raise ValueError('Cannot take a larger sample than population when replace is False')     # numpy's own error
    # np.random.choice(n, 4, replace=False) = permutation(n)[:4]: numpy runs a full n-element Fisher-Yates pass (~11,000 generator
    # words at n = 8192) for four numbers.  transforms.draw_choice_heads replays the stream and keeps only what the four heads need
    # (csrc/nprng_heads.hip): identical rows, identical generator state afterwards.
    return transforms.draw_choice_heads(n_points, 4, max_iter)


def draw_hypothesis_ids_fast(n_points, max_iter):
    """Vectorised draw of 4 distinct indices per hypothesis from numpy's global generator (same distribution as the
    reference's per-iteration np.random.choice, ~1000x faster, but NOT the same random stream)."""
    ids = np.random.randint(0, n_points, size=(max_iter, 4))
    while True:
        s = np.sort(ids, axis=1)
        dup = (s[:, 1:] == s[:, :-1]).any(axis=1)
        if not dup.any():
            return ids.astype(np.int32)
        ids[dup] = np.random.randint(0, n_points, size=(int(dup.sum()), 4))


def voxel_down_sample_device(pts, voxel_size):
    """open3d's PointCloud.voxel_down_sample (aligning.py:65,69) on a (N,3) float64 device tensor: voxel index =
    floor((p - (min_bound - voxel/2)) / voxel), one output point per occupied voxel = the mean of its points (sorted by voxel index;
    open3d returns them in hash-map order, which the caller -- a nearest-neighbour query -- cannot observe)."""
    if pts.shape[0] == 0:
        return pts.clone()
    mn = pts.min(dim=0).values - 0.5 * voxel_size
    idx = torch.floor((pts - mn) / voxel_size).long()
    ext = idx.max(dim=0).values + 1
    if float(ext[0]) * float(ext[1]) * float(ext[2]) >= 2.0 ** 62:
        raise ValueError('voxel_down_sample: the voxel lattice of this cloud does not fit 62 bits (voxel size too small)')
    key = (idx[:, 0] * ext[1] + idx[:, 1]) * ext[2] + idx[:, 2]
    uniq, inv = torch.unique(key, return_inverse=True)
    sums = torch.zeros((uniq.shape[0], 3), dtype=pts.dtype, device=pts.device).index_add_(0, inv, pts)
    return sums / torch.bincount(inv, minlength=uniq.shape[0]).to(pts.dtype).unsqueeze(1)


def _nn_dist(query, ref):
    """cKDTree(ref).query(query)[0] on the device: brute-force float64 nearest neighbour (cg_nearest_neighbor) + the distance."""
    idx = torch.empty((query.shape[0],), dtype=torch.int32, device=query.device)
    check(L.lib().cg_nearest_neighbor(_p(query), ctypes.c_long(query.shape[0]), _p(ref), _c_int(ref.shape[0]), _p(idx), _stream()), 'cg_nearest_neighbor')
    return torch.linalg.vector_norm(query - ref[idx.long()], dim=1)


def _kdtree_eval(d_src, d_dst, transforms, accepted, thres, resolution):
    """The `use_kdtree_for_eval=True` scoring of aligning.py:63-76 for the hypotheses that passed the scale / SVD / extent gates
    (cg_ransac_9d), then the arg-max of aligning.py:112-115: errs = [NN distance of every transformed source point to the
    voxel-down-sampled target, NN distance of every target point to the voxel-down-sampled transformed source]; ratio = share of errs
    <= PassThreshold; inliers = source points whose first distance passes.  One pair of nearest-neighbour launches per accepted
    hypothesis (the reference pipeline itself evaluates directly, predicter.py:162: this branch is built for completeness, not speed)."""
    tgt_ds = voxel_down_sample_device(d_dst, resolution).contiguous()
    best_ratio, best = -1.0, None
    n2 = 2 * d_src.shape[0]
    for h in accepted:
        T = transforms[int(h)].view(4, 4)
        st = (d_src @ T[:3, :3].T + T[:3, 3]).contiguous()
        d1 = _nn_dist(st, tgt_ds)
        d2 = _nn_dist(d_dst, voxel_down_sample_device(st, resolution).contiguous())
        ratio = (int((d1 <= thres).sum()) + int((d2 <= thres).sum())) / n2
        if ratio > best_ratio:                        # np.argmax: the first maximum
            best_ratio, best = ratio, (T.cpu().numpy().copy(), torch.nonzero(d1 <= thres).reshape(-1).cpu().numpy())
    return best


def estimate9DTransform(source, target, PassThreshold, max_iter=1000, use_kdtree_for_eval=False, kdtree_eval_resolution=None,
                        max_scale=np.array([99, 99, 99]), min_scale=np.array([0, 0, 0]), max_dimensions=None, ids=None, device=None,
                        sampling='reference'):
    """-> (4x4 float64 transform, inlier index array) or (None, None), like aligning.py:83-119.
    `ids` (max_iter,4): explicit hypothesis samples; otherwise sampling='reference' reproduces the reference's
    numpy-global-RNG draw call by call (each draw is a full-cloud shuffle in numpy's stream; replayed in C, a few us per hypothesis),
    sampling='fast' draws the same distribution vectorised.
    use_kdtree_for_eval=True (aligning.py:63-76; the live pipeline passes False) scores every accepted hypothesis by two-sided
    nearest-neighbour distances against voxel-down-sampled clouds of `kdtree_eval_resolution` (see _kdtree_eval)."""
    if use_kdtree_for_eval and (kdtree_eval_resolution is None or not kdtree_eval_resolution > 0):
        raise ValueError('use_kdtree_for_eval=True needs kdtree_eval_resolution > 0 (the voxel size of aligning.py:65,69)')
    if device is None:
        if not torch.cuda.is_available():
            raise L.CatgraspAmdError('catgrasp_amd.aligning needs a HIP device (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())
    src = np.ascontiguousarray(np.asarray(source, dtype=np.float64).reshape(-1, 3))
    dst = np.ascontiguousarray(np.asarray(target, dtype=np.float64).reshape(-1, 3))
    assert src.shape == dst.shape
    N = len(src)
    if ids is None:
        ids = draw_hypothesis_ids(N, max_iter) if sampling == 'reference' else draw_hypothesis_ids_fast(N, max_iter)
    ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 4)
    H = len(ids)
    if H == 0:
        return None, None
    d_src = torch.from_numpy(src).to(device); d_dst = torch.from_numpy(dst).to(device)
    d_ids = torch.from_numpy(ids).to(device)
    counts = torch.empty((H,), dtype=torch.int32, device=device)
    transforms = torch.empty((H, 16), dtype=torch.float64, device=device)
    D3 = ctypes.c_double * 3
    mn = D3(*[float(v) for v in np.asarray(min_scale, dtype=np.float64).reshape(3)])
    mx = D3(*[float(v) for v in np.asarray(max_scale, dtype=np.float64).reshape(3)])
    md = D3(*[float(v) for v in np.asarray(max_dimensions, dtype=np.float64).reshape(3)]) if max_dimensions is not None else None
    check(L.lib().cg_ransac_9d(_p(d_src), _p(d_dst), _c_int(N), _p(d_ids), _c_int(H), ctypes.c_double(float(PassThreshold)), mn, mx, md,
                               _p(counts), _p(transforms), _stream()), 'cg_ransac_9d')
    c = counts.cpu().numpy()
    valid = c >= 0
    if not valid.any():
        return None, None
    if use_kdtree_for_eval:
        return _kdtree_eval(d_src, d_dst, transforms, np.flatnonzero(valid), float(PassThreshold), float(kdtree_eval_resolution))
    # ratios = count/N over the accepted hypotheses, arg-max = first maximum (aligning.py:112)
    best = int(np.flatnonzero(valid)[np.argmax(c[valid])])
    mask = torch.empty((N,), dtype=torch.uint8, device=device)
    check(L.lib().cg_similarity_inliers(_p(d_src), _p(d_dst), _c_int(N), _p(transforms[best]), ctypes.c_double(float(PassThreshold)), _p(mask),
                                        _stream()), 'cg_similarity_inliers')
    T = transforms[best].cpu().numpy().reshape(4, 4).copy()
    return T, np.where(mask.cpu().numpy() > 0)[0]
