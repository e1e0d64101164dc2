"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of the per-sample host transforms and
predicter glue of the reference (their originals import open3d/trimesh/autolab_core, which are
not installable here, so they cannot be imported).

The reference draws resample indices from numpy's global RNG (`np.random.choice`,
dataset_grasp.py:73, dataset_nunocs.py:44).  Here the indices are an explicit argument so
the same `ids` feed both the oracle and the HIP path; `draw_ids` reproduces the draw.

Pinned against the reference itself: tests/golden/make_golden_host.py runs the REAL GraspDataset.transform /
NunocsIsolatedDataset.transform / NormalizeCloud / to_homo under inert import stubs (host_golden.npz), and
make_golden_predicter.py the REAL GraspPredicter.predict_batch and the NUNOCS decode (predicter_golden.npz);
tests/test_oracle_host_golden.py checks this file against both.
"""
import numpy as np


def to_homo(pts):
    """Utils.py:396-402."""
    assert pts.ndim == 2
    return np.concatenate((pts, np.ones((pts.shape[0], 1))), axis=-1)


def draw_ids(n_valid, n_pts, rng=np.random):
    """dataset_grasp.py:72-73 / dataset_nunocs.py:43-44: choice with replacement iff too few points."""
    replace = n_valid < n_pts
    return rng.choice(np.arange(n_valid), size=(n_pts), replace=replace)


def grasp_transform(cloud_xyz, cloud_normal, grasp_pose, ids, mean=None, std=None):
    """GraspDataset.transform, phase='test' (dataset_grasp.py:63-91).  float64 in, float64 out.

    Returns dict(input (n_pts,6), cloud_xyz_original (n_pts,3)).  `ids` index the z>=0.1
    filtered cloud, exactly where the reference applies them.
    """
    valid = cloud_xyz[:, 2] >= 0.1                                         # :64
    xyz = cloud_xyz[valid].reshape(-1, 3)
    nrm = cloud_normal[valid].reshape(-1, 3)
    xyz = (np.linalg.inv(grasp_pose) @ to_homo(xyz).T).T[:, :3]           # :69
    nrm = (np.linalg.inv(grasp_pose[:3, :3]) @ nrm.T).T                    # :70
    xyz = xyz[ids]                                                         # :74
    nrm = nrm[ids].reshape(-1, 3)
    inp = np.concatenate((xyz, nrm), axis=-1)                              # :82
    if mean is not None:
        inp = (inp - mean.reshape(1, -1)) / (std.reshape(1, -1) + 1e-15)   # :84-85
    return {'input': inp, 'cloud_xyz_original': xyz.copy()}


def normalize_cloud(xyz):
    """augmentations.NormalizeCloud (augmentations.py:66-75): isotropic min/max-extent scale."""
    max_xyz = xyz.max(axis=0)
    min_xyz = xyz.min(axis=0)
    scale = (max_xyz - min_xyz).max()
    return (xyz - min_xyz) / (scale + 1e-15)


def nunocs_transform(cloud_xyz, cloud_normal, ids, mean=None, std=None):
    """NunocsIsolatedDataset.transform, phase='test' (dataset_nunocs.py:38-65)."""
    keep_ids = np.arange(cloud_xyz.shape[0])
    valid = cloud_xyz[:, 2] >= 0.1                                         # :40
    keep_ids = keep_ids[valid]
    xyz = cloud_xyz[valid]
    xyz = xyz[ids]                                                         # :45
    keep_ids = keep_ids[ids]
    nrm = cloud_normal[keep_ids].reshape(-1, 3)                            # :49
    xyz_original = xyz.copy()
    xyz_n = normalize_cloud(xyz)                                           # :56
    inp = np.concatenate((xyz_n, nrm), axis=-1)                            # :57
    if mean is not None:
        inp = (inp - mean.reshape(1, -1)) / (std.reshape(1, -1) + 1e-15)   # :59-60
    return {'input': inp, 'cloud_xyz_original': xyz_original, 'keep_ids': keep_ids}


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def predict_batch_post(logits):
    """predicter.py:86-91: softmax over classes; per row [argmax, confidence, probs]."""
    pred = softmax(np.asarray(logits, dtype=np.float32).astype(np.float64), axis=1).astype(np.float32)
    out = []
    for b in range(len(pred)):
        cur = pred[b]
        lab = cur.argmax()
        out.append([lab, cur[lab], cur])
    return out


def p_G(probs, n_classes):
    """run_grasp_simulation.py:313: expected bin index / n_classes."""
    probs = np.asarray(probs)
    return (probs * np.arange(probs.shape[-1])).sum(axis=-1) / n_classes


def nunocs_decode(logits, n_bins):
    """predicter.py:144-150: per-axis argmax bin -> coordinate in [-0.5, 0.5); z-axis confidence."""
    pred = np.asarray(logits).reshape(-1, 3, n_bins)
    coords = pred.argmax(axis=-1).astype(np.float32) * np.float32(1.0 / n_bins)
    probs = softmax(pred.astype(np.float64), axis=-1)
    zbin = pred[:, 2, :].argmax(axis=-1)
    conf_z = probs[np.arange(len(pred)), 2, zbin].astype(np.float32)
    return coords - 0.5, conf_z
