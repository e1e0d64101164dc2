"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of my_cpp directionVecToRotation / augmentGraspPoses
(my_cpp/common.cpp:75-153), float32 loop counters, SVD orthonormalisation (R = U V^T) like Eigen::JacobiSVD.

PINNED to the reference's own C++: my_cpp as a whole cannot be built here, but these two functions need Eigen alone (vendored in
the reference tree), so oracle/build_ref.py:build_augment compiles them from the lines where they lie into
oracle/_ref/libaugment_ref.so; tests/golden/make_golden_augment.py runs that binary and commits tests/golden/augment_golden.npz
(207 directions incl. the degenerate ones, 4 augmentGraspPoses calls = 1104 poses), against which this restatement
(tests/test_oracle_host_golden.py, <= 1e-6) and the HIP kernel (tests/test_collision_gpu.py, <= 1e-5) are checked.  The python twin
Utils.directionVecToRotation additionally pins `direction_vec_to_rotation` in tests/golden/host_golden.npz."""
import numpy as np


def _svd_rot(R):
    u, _, vt = np.linalg.svd(R.astype(np.float64))
    return u @ vt


def direction_vec_to_rotation(direction, ref):
    """common.cpp:75-115."""
    d = np.asarray(direction, dtype=np.float64); d = d / np.linalg.norm(d)
    ref = np.asarray(ref, dtype=np.float64)
    v = np.cross(d, ref)
    if np.linalg.norm(v) < 1e-5:
        return np.eye(3)
    s = np.linalg.norm(v); c = d @ ref
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    R = np.eye(3) + K + K @ K * (1 - c) / (s * s)
    return _svd_rot(R.T)


def augment_grasp_poses(R0, selected_point, sphere_pts, inplane_rot_step, hand_depth, approach_step, init_bite):
    """common.cpp:118-153 restricted to the valid rows of sphere_pts (the reference's `i<sphere_pts.size()` reads
    past the end of the matrix)."""
    R0 = np.asarray(R0, dtype=np.float64); p = np.asarray(selected_point, dtype=np.float64)
    Rs = [R0]
    for sp in np.asarray(sphere_pts, dtype=np.float64).reshape(-1, 3):
        R_sphere = direction_vec_to_rotation(sp, [1, 0, 0])
        x_rot = np.float32(0)
        while x_rot < np.float32(180):
            a = float(x_rot) / 180.0 * np.pi
            Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
            Rs.append(R0 @ R_sphere @ Rx)
            x_rot = np.float32(x_rot + np.float32(inplane_rot_step))
    poses = []
    for R in Rs:
        R = _svd_rot(R)
        a = R[:, 0]
        d = np.float32(0)
        while d < np.float32(hand_depth):
            T = np.eye(4); T[:3, :3] = R; T[:3, 3] = p + init_bite * a + a * float(d)
            poses.append(T)
            d = np.float32(d + np.float32(approach_step))
    return np.array(poses)
