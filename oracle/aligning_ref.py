"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of aligning.estimate9DTransform (aligning.py:23-119).
PARITY UNPINNED w.r.t. OpenCV: cv2 is not installable here.  `cv2.estimateAffine3D` on exactly 4 correspondences is
restated as the exact affine map through them (its RANSAC has a single possible sample and its refit uses the same 4
points); a degenerate (coplanar) sample yields no model.  Everything after that call is pinned against the reference
itself: tests/golden/make_golden_host.py runs the REAL aligning.estimate9DTransform_worker with that one substitution
(host_golden.npz, tests/test_oracle_host_golden.py).
The `use_kdtree_for_eval=True` branch (aligning.py:63-76) additionally routes both clouds through open3d's
PointCloud.voxel_down_sample -- open3d is not installable here either, so `voxel_down_sample` below restates it from general
knowledge of Open3D (PARITY UNPINNED w.r.t. open3d: voxel index = floor((p - (min_bound - voxel/2)) / voxel), one output point per
occupied voxel = the mean of its points; the output ORDER is open3d's hash-map order, which this branch never observes: it only
takes nearest-neighbour distances to the down-sampled set).  The rest of the branch -- scipy.cKDTree queries (scipy IS here), the
two-sided error vector, ratio and inliers -- is pinned by tests/golden/make_golden_aligning_kd.py, which runs the REAL worker with
that stand-in (aligning_kd_golden.npz)."""
import numpy as np


def voxel_down_sample(points, voxel_size):
    """open3d.geometry.PointCloud.voxel_down_sample restated: -> (M,3) float64 voxel centroids, sorted by voxel index."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(pts) == 0:
        return pts.copy()
    mn = pts.min(axis=0) - voxel_size * 0.5
    idx = np.floor((pts - mn) / voxel_size).astype(np.int64)
    uniq, inv = np.unique(idx, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    sums = np.zeros((len(uniq), 3)); np.add.at(sums, inv, pts)
    return sums / np.bincount(inv, minlength=len(uniq)).reshape(-1, 1)


def affine_from_4(src4, dst4):
    M = np.concatenate([src4, np.ones((4, 1))], axis=1)
    if abs(np.linalg.det(M)) < 1e-13 * max(np.abs(src4).max(), 1e-300) ** 3:
        return None
    X = np.linalg.solve(M, dst4)             # (4,3): dst = [src 1] X
    T = np.eye(4); T[:3, :3] = X[:3].T; T[:3, 3] = X[3]
    return T


def worker(cur_src, cur_dst, source, target, thres, max_scale, min_scale, max_dimensions, use_kdtree_for_eval=False, kdtree_eval_resolution=None):
    """aligning.py:33-81.  -> None | (inlier count, transform, inliers) for the direct evaluation, (ratio, transform, inliers) for
    the kd-tree one (its error vector has 2N entries: source->target and target->source nearest-neighbour distances)."""
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
    if use_kdtree_for_eval:                      # aligning.py:63-76
        from scipy.spatial import cKDTree
        d1, _ = cKDTree(voxel_down_sample(target, kdtree_eval_resolution)).query(st)
        d2, _ = cKDTree(voxel_down_sample(st, kdtree_eval_resolution)).query(target)
        errs = np.concatenate((d1, d2), axis=0).reshape(-1)
        return float(np.sum(errs <= thres) / len(errs)), new_t, np.where(d1 <= thres)[0]
    errs = np.linalg.norm(st - target, axis=-1)
    return int(np.sum(errs <= thres)), new_t, np.where(errs <= thres)[0]


def estimate9DTransform(source, target, PassThreshold, ids, max_scale, min_scale, max_dimensions=None, use_kdtree_for_eval=False,
                        kdtree_eval_resolution=None):
    outs = []
    for i in range(len(ids)):
        o = worker(source[ids[i]], target[ids[i]], source, target, PassThreshold, np.asarray(max_scale), np.asarray(min_scale), max_dimensions,
                   use_kdtree_for_eval, kdtree_eval_resolution)
        outs.append(o)
    good = [o for o in outs if o is not None]
    if not good:
        return None, None, outs
    best = int(np.argmax([o[0] for o in good]))
    return good[best][1], good[best][2], outs
