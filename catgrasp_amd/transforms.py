"""Host-side preparation for the device input-transform kernels: the parts of
GraspDataset.transform / NunocsIsolatedDataset.transform (dataset_grasp.py:63-91,
dataset_nunocs.py:38-65) that are bookkeeping rather than per-point arithmetic --
the z>=0.1 validity mask, drawing the resample indices, inverting the 4x4 poses in float64 and
re-expressing them for an object-centred float32 cloud.  The per-point arithmetic itself runs in
cg_build_grasp_input / cg_build_nunocs_input.
"""
import numpy as np
import torch


def valid_mask(cloud_xyz):
    """dataset_grasp.py:64 / dataset_nunocs.py:40."""
    return np.asarray(cloud_xyz)[:, 2] >= 0.1


def draw_ids_reference(n_valid, n_pts, count):
    """Resample indices drawn exactly as the reference does: one `np.random.choice` per sample from
    numpy's GLOBAL generator (dataset_grasp.py:72-73), with replacement iff n_valid < n_pts.
    Seeding numpy therefore reproduces the reference's draws.  -> (count, n_pts) int32."""
    replace = n_valid < n_pts
    base = np.arange(n_valid)
    out = np.empty((count, n_pts), dtype=np.int32)
    for i in range(count):
        out[i] = np.random.choice(base, size=(n_pts), replace=replace)
    return out


def draw_ids_device(n_valid, n_pts, count, device, generator=None):
    """Statistically equivalent draw on the device (NOT numpy's stream): a random permutation prefix per
    row when n_valid >= n_pts, iid uniform indices otherwise.  -> (count, n_pts) int32 cuda tensor."""
    if n_valid < n_pts:
        return torch.randint(0, n_valid, (count, n_pts), device=device, generator=generator, dtype=torch.int32)
    keys = torch.rand((count, n_valid), device=device, generator=generator)
    return keys.argsort(dim=1)[:, :n_pts].to(torch.int32).contiguous()


class DeviceCloud:
    """An object cloud resident in HBM: z-filtered, centred on its centroid in float64 and rounded once to
    float32 (camera-frame coordinates are ~0.6 m with ~1 cm extent: centring keeps 3 more decimal digits)."""

    def __init__(self, cloud_xyz, cloud_normal, device):
        xyz = np.asarray(cloud_xyz, dtype=np.float64)
        nrm = np.asarray(cloud_normal, dtype=np.float64)
        m = valid_mask(xyz)
        self.keep_ids = np.arange(len(xyz))[m]
        self.xyz64 = xyz[m].reshape(-1, 3)
        self.normal64 = nrm[m].reshape(-1, 3)
        self.n = len(self.xyz64)
        self.center = self.xyz64.mean(axis=0) if self.n else np.zeros(3)
        self.xyz = torch.from_numpy((self.xyz64 - self.center).astype(np.float32)).to(device)
        self.normal = torch.from_numpy(self.normal64.astype(np.float32)).to(device)
        self.device = device


def pose_inverse_rows(grasp_poses, center):
    """inv(grasp_pose) in float64 (dataset_grasp.py:69-70 use np.linalg.inv), re-expressed for a cloud
    shifted by -center, rounded to float32: rows (G,12) of [R | t] with x_grasp = R x_centred + t."""
    P = np.asarray(grasp_poses, dtype=np.float64).reshape(-1, 4, 4)
    if len(P) == 0:
        return np.zeros((0, 12), dtype=np.float32)
    Pinv = np.linalg.inv(P)
    # normals use inv(R) of the rotation block alone (dataset_grasp.py:70); for a valid pose (last row 0 0 0 1)
    # that is the upper-left block of inv(P).
    R = Pinv[:, :3, :3]
    t = Pinv[:, :3, 3] + R @ np.asarray(center, dtype=np.float64)
    return np.concatenate([R, t[:, :, None]], axis=2).reshape(-1, 12).astype(np.float32)


def normalizer_device(cfg, device):
    """(mean, 1/(std+1e-15)) float32 device tensors from cfg['mean'], cfg['std'] (dataset_grasp.py:84-85), or (None, None)."""
    if 'mean' not in cfg:
        return None, None
    mean = np.asarray(cfg['mean'], dtype=np.float64).reshape(-1)
    std = np.asarray(cfg['std'], dtype=np.float64).reshape(-1)
    return (torch.from_numpy(mean.astype(np.float32)).to(device),
            torch.from_numpy((1.0 / (std + 1e-15)).astype(np.float32)).to(device))


def get_symmetry_tfs(class_name):
    """Symmetry transforms of an object category in its canonical frame (Utils.py:79-94 with allow_reflection=True; the
    reference builds them with transformations.euler_matrix(x, 0, z, 'sxyz') = Rz(z) Rx(x)): nut 2 x 6 = 12, hnm 2, screw 72."""
    def rz(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])

    def rx(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])
    if class_name == 'nut':
        return [rz(z) @ rx(x) for x in np.arange(0, 360, 180) / 180 * np.pi for z in np.arange(0, 360, 60) / 180 * np.pi]
    if class_name == 'hnm':
        return [rz(z) for z in (0, np.pi)]
    if class_name == 'screw':
        return [rz(z) for z in np.arange(0, 360, 5) / 180.0 * np.pi]
    raise RuntimeError(f'{class_name} not found')
