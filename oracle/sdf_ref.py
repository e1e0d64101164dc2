"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of meshpy Sdf3D lookups
(meshpy/meshpy/sdf.py:312-343,351-357,377-389) and the SdfFile text reader (sdf_file.py:59-87).
The originals import open3d / autolab_core; tests/golden/make_golden_host.py imports them under inert stubs and runs the REAL
Sdf3D._signed_distance / _signed_distance_batch / is_any_points_inside / SdfFile._read_3d, which pins this restatement
(host_golden.npz, tests/test_oracle_host_golden.py)."""
import numpy as np

MIN_X, MAX_X = [0, 2, 3, 5], [1, 4, 6, 7]     # sdf.py:219-224
MIN_Y, MAX_Y = [0, 1, 3, 6], [2, 4, 5, 7]
MIN_Z, MAX_Z = [0, 1, 2, 4], [3, 5, 6, 7]


def signed_distance(data, coords, fast=False):
    """sdf.py:312-343.  data (nx,ny,nz), coords (3,N) in grid units (float64)."""
    dims = np.array(data.shape)
    coords = np.array(coords, dtype=np.float64).reshape(3, -1)
    for i in range(3):
        coords[i] = np.clip(coords[i], 0, dims[i] - 1)
    if fast:
        c = coords.round().astype(int)
        return data[c[0], c[1], c[2]]
    mn = np.floor(coords)
    mx = mn + 1
    corners = np.zeros((coords.shape[1], 8, 3))
    corners[:, MIN_X, 0] = mn[0].reshape(-1, 1); corners[:, MAX_X, 0] = mx[0].reshape(-1, 1)
    corners[:, MIN_Y, 1] = mn[1].reshape(-1, 1); corners[:, MAX_Y, 1] = mx[1].reshape(-1, 1)
    corners[:, MIN_Z, 2] = mn[2].reshape(-1, 1); corners[:, MAX_Z, 2] = mx[2].reshape(-1, 1)
    sd = np.zeros(coords.shape[1])
    corners = corners.astype(int)
    for i in range(8):
        cur = corners[:, i]
        oob = (cur < 0).any(axis=1) | (cur >= dims.reshape(1, 3)).any(axis=1)
        inb = ~oob
        vals = np.zeros(len(cur))
        vals[inb] = data[cur[inb, 0], cur[inb, 1], cur[inb, 2]]
        w = np.prod(1 - np.abs(cur - coords.T), axis=1)
        sd = sd + w * vals
    return sd


def signed_distance_batch(data, coords):
    """sdf.py:345-357: round (half-even) -> clamp -> gather.  coords (B,3,N)."""
    dims = data.shape
    c = np.round(np.asarray(coords, dtype=np.float32)).astype(np.int64)
    for a in range(3):
        c[:, a] = np.clip(c[:, a], 0, dims[a] - 1)
    return data[c[:, 0], c[:, 1], c[:, 2]]


def is_any_points_inside(data, coords):
    """sdf.py:377-389."""
    c = np.asarray(coords).round().astype(int)
    for a in range(3):
        c = c[:, c[a] >= 0]
        c = c[:, c[a] < data.shape[a]]
    return bool((data[c[0], c[1], c[2]] < 0).any())


def read_sdf_file(path):
    """sdf_file.py:59-87 (the reference's python triple loop)."""
    with open(path, 'r') as f:
        nx, ny, nz = [int(i) for i in f.readline().split()]
        origin = np.array([float(i) for i in f.readline().split()])
        resolution = float(f.readline())
        data = np.zeros((nx, ny, nz))
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    data[i][j][k] = float(f.readline())
    return data, origin, resolution


def box_sdf_grid(lo, hi, resolution=0.001, padding=5):
    """Analytic signed distance of an axis-aligned box on the SDFGen lattice (make_sdf.py:30-34):
    dim = ceil(max_extent/res) + 2*padding, cubic grid, origin = lower corner - padding*res."""
    lo = np.asarray(lo, float); hi = np.asarray(hi, float)
    dim = int(np.ceil((hi - lo).max() / resolution) + 2 * padding)
    origin = lo - padding * resolution
    g = origin[:, None] + resolution * np.arange(dim)[None, :]
    X, Y, Z = np.meshgrid(g[0], g[1], g[2], indexing='ij')
    P = np.stack([X, Y, Z], -1)
    c = (lo + hi) / 2; h = (hi - lo) / 2
    q = np.abs(P - c) - h
    outside = np.linalg.norm(np.maximum(q, 0), axis=-1)
    inside = np.minimum(q.max(axis=-1), 0)
    return (outside + inside), origin, resolution
