"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy/scipy restatement of compute_grasp_affordance_worker
(run_grasp_simulation.py:50-75) and get_finger_contact_area (pybullet_env/env_grasp.py:243-283).

Pinned against the reference itself: tests/golden/make_golden_affordance.py runs the REAL get_finger_contact_area (open3d
replaced by a small functional stand-in) -> affordance_golden.npz, checked in tests/test_oracle_host_golden.py."""
import numpy as np


def to_homo(p):
    return np.concatenate([p, np.ones((len(p), 1))], axis=1)


def get_finger_contact_area(finger_vertices, ob_in_finger, ob_pts, grip_dir, ob_normals, surface_tol):
    grip_dir = np.array(grip_dir, dtype=float); grip_dir = grip_dir / np.linalg.norm(grip_dir)
    cur = (ob_in_finger @ to_homo(ob_pts).T).T[:, :3]
    cur_n = (ob_in_finger[:3, :3] @ ob_normals.T).T
    V = finger_vertices
    within = (cur[:, 0] >= V[:, 0].min()) & (cur[:, 0] <= V[:, 0].max()) & (cur[:, 2] >= V[:, 2].min()) & (cur[:, 2] <= V[:, 2].max())
    if within.sum() == 0:
        return None
    wp = cur[within]; wn = cur_n[within]
    if np.allclose(grip_dir, [0, 1, 0]):
        dist = np.abs(wp[:, 1] - wp[:, 1].min())
    elif np.allclose(grip_dir, [0, -1, 0]):
        dist = np.abs(wp[:, 1] - wp[:, 1].max())
    else:
        raise RuntimeError
    cm = dist <= surface_tol
    if cm.sum() == 0:
        return None
    dist = dist[cm]; sp = wp[cm]; sn = wn[cm]
    cn = sn[np.abs(dist).argmin()].copy(); cn /= np.linalg.norm(cn)
    if np.dot(cn, grip_dir) > 0:
        return None
    return (np.linalg.inv(ob_in_finger) @ to_homo(sp).T).T[:, :3]


def grasp_affordance(grasp_in_cam, finger_mesh_in_grasp, pts, normals, canonical_affordance, kdtree, grip_dirs, finger_vertices, surface_tol=0.005):
    cam_in_finger = np.linalg.inv(finger_mesh_in_grasp) @ np.linalg.inv(grasp_in_cam)
    vals, counts = [], []
    for i in range(len(grip_dirs)):
        sp = get_finger_contact_area(finger_vertices[i], cam_in_finger, pts, grip_dirs[i], normals, surface_tol)
        if sp is None:
            counts.append(0); continue
        counts.append(len(sp))
        _, idx = kdtree.query(sp)
        vals.append(canonical_affordance[idx].mean())
    return (np.array(vals).mean() if vals else np.nan), counts
