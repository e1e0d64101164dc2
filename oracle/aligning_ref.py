"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of aligning.estimate9DTransform (aligning.py:23-119).
PARITY UNPINNED w.r.t. OpenCV: cv2 is not installable here.  `cv2.estimateAffine3D` on exactly 4 correspondences is
restated as the exact affine map through them (its RANSAC has a single possible sample and its refit uses the same 4
points); a degenerate (coplanar) sample yields no model.  Everything after that call is pinned against the reference
itself: tests/golden/make_golden_host.py runs the REAL aligning.estimate9DTransform_worker with that one substitution
(host_golden.npz, tests/test_oracle_host_golden.py)."""
import numpy as np


def affine_from_4(src4, dst4):
    M = np.concatenate([src4, np.ones((4, 1))], axis=1)
    if abs(np.linalg.det(M)) < 1e-13 * max(np.abs(src4).max(), 1e-300) ** 3:
        return None
    X = np.linalg.solve(M, dst4)             # (4,3): dst = [src 1] X
    T = np.eye(4); T[:3, :3] = X[:3].T; T[:3, 3] = X[3]
    return T


def worker(cur_src, cur_dst, source, target, thres, max_scale, min_scale, max_dimensions):
    """aligning.py:33-81 (use_kdtree_for_eval=False)."""
    transform = affine_from_4(cur_src, cur_dst)
    if transform is None:
        return None
    scales = np.linalg.norm(transform[:3, :3], axis=0)
    if (scales > max_scale).any() or (scales < min_scale).any() or not (scales > 0).all():
        return None
    R = transform[:3, :3] / scales.reshape(1, 3)
    u, s, vh = np.linalg.svd(R)
    if s.min() < 0.8 or s.max() > 1.2:
        return None
    R = u @ vh
    if np.linalg.det(R) < 0:
        return None
    new_t = transform.copy(); new_t[:3, :3] = R @ np.diag(scales)
    if max_dimensions is not None:
        can = (np.linalg.inv(new_t) @ np.concatenate([target, np.ones((len(target), 1))], 1).T).T[:, :3]
        if ((can.max(axis=0) - can.min(axis=0)) > max_dimensions).any():
            return None
    st = (new_t @ np.concatenate([source, np.ones((len(source), 1))], 1).T).T[:, :3]
    errs = np.linalg.norm(st - target, axis=-1)
    return int(np.sum(errs <= thres)), new_t, np.where(errs <= thres)[0]


def estimate9DTransform(source, target, PassThreshold, ids, max_scale, min_scale, max_dimensions=None):
    outs = []
    for i in range(len(ids)):
        o = worker(source[ids[i]], target[ids[i]], source, target, PassThreshold, np.asarray(max_scale), np.asarray(min_scale), max_dimensions)
        outs.append(o)
    good = [o for o in outs if o is not None]
    if not good:
        return None, None, outs
    best = int(np.argmax([o[0] for o in good]))
    return good[best][1], good[best][2], outs
