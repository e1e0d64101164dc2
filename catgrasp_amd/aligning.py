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
    # np.random.choice(n, 4, replace=False) = permutation(n)[:4]: a full n-element shuffle per hypothesis.  Replayed in C from numpy's
    # own generator state (transforms.NumpyChoiceStream: identical rows, identical state afterwards), ~2.5x faster than the python loop.
    return transforms.draw_ids_reference(n_points, 4, max_iter)


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


def estimate9DTransform(source, target, PassThreshold, max_iter=1000, use_kdtree_for_eval=False, kdtree_eval_resolution=None,
                        max_scale=np.array([99, 99, 99]), min_scale=np.array([0, 0, 0]), max_dimensions=None, ids=None, device=None,
                        sampling='reference'):
    """-> (4x4 float64 transform, inlier index array) or (None, None), like aligning.py:83-119.
    `ids` (max_iter,4): explicit hypothesis samples; otherwise sampling='reference' reproduces the reference's
    numpy-global-RNG draw call by call (each draw is a full-cloud shuffle in numpy's stream: ~50 us per hypothesis even replayed in C),
    sampling='fast' draws the same distribution vectorised."""
    if use_kdtree_for_eval:
        raise NotImplementedError('use_kdtree_for_eval=True is not built (the reference pipeline passes False, predicter.py:162)')
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
    # ratios = count/N over the accepted hypotheses, arg-max = first maximum (aligning.py:112)
    best = int(np.flatnonzero(valid)[np.argmax(c[valid])])
    mask = torch.empty((N,), dtype=torch.uint8, device=device)
    check(L.lib().cg_similarity_inliers(_p(d_src), _p(d_dst), _c_int(N), _p(transforms[best]), ctypes.c_double(float(PassThreshold)), _p(mask),
                                        _stream()), 'cg_similarity_inliers')
    T = transforms[best].cpu().numpy().reshape(4, 4).copy()
    return T, np.where(mask.cpu().numpy() > 0)[0]
