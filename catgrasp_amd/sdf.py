"""Device-resident counterpart of meshpy's `Sdf3D` / `SdfFile` (meshpy/meshpy/sdf.py:216-389,
meshpy/meshpy/sdf_file.py:59-87) for the lookups the grasp path prepares: grid<->object transform,
trilinear / nearest signed distance, any-point-inside, and the batched per-candidate form.
Lookups run in csrc/sdf.hip; there is no CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int
_c_long = ctypes.c_long


class Sdf3D:
    def __init__(self, sdf_data, origin, resolution, device=None):
        data = np.asarray(sdf_data)
        if data.ndim != 3:
            raise ValueError(f'sdf data shape wrong: {data.shape}')
        if device is None:
            if not torch.cuda.is_available():
                raise L.CatgraspAmdError('catgrasp_amd.sdf needs a HIP device (no CPU fallback)')
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.dims_ = np.array(data.shape)
        self.origin_ = np.asarray(origin, dtype=np.float64).reshape(3)
        self.resolution_ = float(resolution)
        self.data_torch = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(self.device)
        # T_world_grid = inverse of SimilarityTransform(translation=origin, scale=resolution) (sdf.py:255-264)
        self.T_world_grid = np.eye(4)
        self.T_world_grid[:3, :3] /= self.resolution_
        self.T_world_grid[:3, 3] = -self.origin_ / self.resolution_

    @property
    def dimensions(self):
        return self.dims_

    @property
    def origin(self):
        return self.origin_

    @property
    def resolution(self):
        return self.resolution_

    # ---- frame conversion (sdf.py:362-373, :659-679) ----
    def transform_pt_obj_to_grid(self, x_sdf):
        """(3,N) object-frame metres -> grid units: (x - origin) / resolution."""
        x = np.asarray(x_sdf, dtype=np.float64)
        return (x - self.origin_.reshape(3, 1)) / self.resolution_

    def _coords(self, coords):
        t = coords if isinstance(coords, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32)
        if t.ndim == 2:
            t = t.unsqueeze(0)
        if t.ndim != 3 or t.shape[1] != 3:
            raise IndexError('Indexing must be 3 dimensional')           # sdf.py:330-331
        return t.contiguous()

    def _lookup(self, coords, mode):
        t = self._coords(coords)
        B, _, N = t.shape
        out = torch.empty((B, N), dtype=torch.float32, device=self.device)
        nx, ny, nz = (int(v) for v in self.dims_)
        check(L.lib().cg_sdf_lookup(_p(self.data_torch), _c_int(nx), _c_int(ny), _c_int(nz), _p(t), _c_long(B), _c_long(N),
                                    _c_int(mode), _p(out), _stream()), 'cg_sdf_lookup')
        return out

    def _signed_distance(self, coords, fast=False):
        """sdf.py:312-343.  coords (3,N) grid units -> (N,) tensor on the device."""
        return self._lookup(coords, 1 if fast else 0)[0]

    def _signed_distance_batch(self, coords, fast=True):
        """sdf.py:345-360.  coords (B,3,N) -> (B,N)."""
        if not fast:
            raise NotImplementedError                                      # as the reference (sdf.py:359-360)
        return self._lookup(coords, 1)

    def is_any_points_inside(self, coords):
        """sdf.py:377-389."""
        t = self._coords(coords)
        assert t.shape[0] == 1
        flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        nx, ny, nz = (int(v) for v in self.dims_)
        check(L.lib().cg_sdf_any_inside(_p(self.data_torch), _c_int(nx), _c_int(ny), _c_int(nz), _p(t), _c_long(t.shape[2]), _p(flag),
                                        _stream()), 'cg_sdf_any_inside')
        return bool(flag.item())

    def is_any_points_inside_batch(self, sdf_in_cam, pts_cam):
        """For E poses of the SDF's object frame in the camera frame (E,4,4) and camera-frame points (P,3):
        out[e] = is_any_points_inside(T_world_grid . inv(sdf_in_cam[e]) . pts).  Returns a bool tensor (E,)."""
        poses = np.asarray(sdf_in_cam, dtype=np.float64).reshape(-1, 4, 4)
        E = len(poses)
        xf = (self.T_world_grid[None] @ np.linalg.inv(poses))[:, :3, :].reshape(E, 12).astype(np.float32) if E else np.zeros((0, 12), np.float32)
        xf_d = torch.from_numpy(np.ascontiguousarray(xf)).to(self.device)
        pts = pts_cam if isinstance(pts_cam, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pts_cam, dtype=np.float32))
        pts = pts.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.zeros((E,), dtype=torch.uint8, device=self.device)
        nx, ny, nz = (int(v) for v in self.dims_)
        check(L.lib().cg_sdf_points_inside_batch(_p(self.data_torch), _c_int(nx), _c_int(ny), _c_int(nz), _p(xf_d), _c_long(E), _p(pts),
                                                 _c_int(pts.shape[0]), _p(out), _stream()), 'cg_sdf_points_inside_batch')
        return out.bool()


class SdfFile:
    """Text .sdf reader (meshpy/meshpy/sdf_file.py:59-87): `nx ny nz` / `ox oy oz` / `resolution` / then
    nx*ny*nz values with x fastest and z slowest, stored as data[i][j][k]."""

    def __init__(self, filepath):
        self.filepath_ = filepath

    @property
    def filepath(self):
        return self.filepath_

    def read_arrays(self):
        """-> (data (nx,ny,nz) float64 with data[i][j][k], origin (3,), resolution): the three constructor arguments
        sdf_file.py:87 hands to Sdf3D; None if the file does not exist."""
        import os
        if not os.path.exists(self.filepath_):
            return None
        with open(self.filepath_, 'r') as f:
            nx, ny, nz = [int(i) for i in f.readline().split()]
            origin = np.array([float(i) for i in f.readline().split()])
            resolution = float(f.readline())
            vals = np.array(f.read().split(), dtype=np.float64)
        data = vals[:nx * ny * nz].reshape(nz, ny, nx).transpose(2, 1, 0)   # file order: k slowest, i fastest
        return np.ascontiguousarray(data), origin, resolution

    def read(self, device=None):
        parts = self.read_arrays()
        if parts is None:
            return None
        return Sdf3D(parts[0], parts[1], parts[2], device=device)

    @staticmethod
    def write(path, data, origin, resolution):
        nx, ny, nz = data.shape
        with open(path, 'w') as f:
            f.write(f'{nx} {ny} {nz}\n{origin[0]} {origin[1]} {origin[2]}\n{resolution}\n')
            for v in np.asarray(data).transpose(2, 1, 0).reshape(-1):
                f.write(f'{float(v)}\n')
