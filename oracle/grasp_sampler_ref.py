"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy/scipy restatement of PointConeGraspSampler.sample_one_surface_point and
the centring loop of sample_grasps (dexnet/grasping/grasp_sampler.py:189-198,225-298), Utils.directionVecToRotation /
normalizeRotation (Utils.py:172-178,262-290).

Pinned against the reference itself: tests/golden/make_golden_host.py runs the REAL sample_one_surface_point /
directionVecToRotation / normalizeRotation under import stubs (host_golden.npz, tests/test_oracle_host_golden.py), and
make_golden_sampler.py the REAL class-level sample_grasps (sampler_golden.npz, tests/test_sampler_classes.py)."""
import numpy as np
from scipy.spatial import cKDTree


def normalize_rotation(R):
    return R / np.linalg.norm(R, axis=0).reshape(1, 3)


def direction_vec_to_rotation(direction, ref):
    direction = direction / np.linalg.norm(direction)
    v = np.cross(direction, ref)
    if (v == 0).all():
        return np.eye(3)
    s = np.linalg.norm(v); c = direction.dot(ref)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    R = np.identity(3) + K + K.dot(K) * (1 - c) / (s ** 2)
    return normalize_rotation(R.T)


class ConeSampler:
    def __init__(self, r_ball, hand_depth, init_bite, approach_step):
        self.r_ball = r_ball; self.hand_depth = hand_depth; self.init_bite = init_bite; self.approach_step = approach_step

    def sample_one_surface_point(self, p, n, pts, nrms, sphere_pts, flip_minor=False):
        tree = cKDTree(pts)
        while True:
            idx = np.array(tree.query_ball_point(p.reshape(1, 3), r=self.r_ball)[0]).astype(int).reshape(-1)
            sq = np.linalg.norm(p.reshape(1, 3) - pts[idx], axis=-1) ** 2
            M = np.zeros((3, 3))
            for k in range(len(idx)):
                if sq[k] != 0:
                    nn = nrms[idx[k]].reshape(-1, 1)
                    if np.linalg.norm(nn) != 0:
                        nn = nn / np.linalg.norm(nn)
                    M += nn @ nn.T
            if sum(sum(M)) == 0:
                self.r_ball *= 2            # persistent, as in the reference (:245)
                continue
            break
        a = -n.reshape(3); a = a / np.linalg.norm(a)
        eigval, eigvec = np.linalg.eig(M)
        self.last_eigvals = np.sort(np.real(eigval))
        minor = np.real(eigvec[:, np.argmin(eigval)]).reshape(3)
        if flip_minor:
            minor = -minor                  # eigenvector sign is LAPACK's choice; both are valid reference outputs
        minor = minor - np.dot(a, minor) / np.dot(a, a) * a
        minor /= np.linalg.norm(minor)
        major = np.cross(minor, a); major = major / np.linalg.norm(major)
        R0 = np.stack([a, major, minor], axis=1)
        Rs = [R0]
        for sp in sphere_pts:
            Rsph = direction_vec_to_rotation(sp.copy(), np.array([1., 0, 0]))
            for x_rot in np.arange(0, 180, 30):
                ang = x_rot * np.pi / 180
                Rx = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
                Rs.append(R0 @ Rsph @ Rx)
        poses = []
        for R in Rs:
            R = normalize_rotation(R)
            ad = R[:, 0]
            for d in np.arange(0, self.hand_depth, self.approach_step):
                T = np.eye(4); T[:3, :3] = R; T[:3, 3] = p + self.init_bite * ad + ad * d
                poses.append(T)
        return np.array(poses)


def center_between_gripper(poses, pts):
    out = []
    for g in poses:
        pig = (np.linalg.inv(g) @ np.concatenate([pts, np.ones((len(pts), 1))], 1).T).T[:, :3]
        c = (pig.max(axis=0) + pig.min(axis=0)) / 2
        off = np.eye(4); off[:3, 3] = [0, c[1], 0]
        out.append(g @ off)
    return np.array(out)
