"""Drop-in for the reference's `my_cpp` pybind module (my_cpp/pybind.cpp:11-23) on MI355X.

Same names, argument order and meaning as the reference:
  * ``CollisionManager`` -- registerMesh / registerPointCloud / setTransform / isAnyCollision
    (my_cpp/collision_manager.h:55-69)
  * ``filterGraspPose(...20 args...)`` -> list of surviving 4x4 float32 grasp_in_cam
    (my_cpp/common.h:60; callers dexnet/grasping/grasp_sampler.py:216,:345)
  * ``augmentGraspPoses`` / ``makeOccupancyGridFromCloudScan`` (my_cpp/common.h:58,61)
plus ``filterGraspPoseDetailed`` (new): the same evaluation in *input order* with a per-evaluation reject
code -- the deterministic "collision mask" the reference's unordered list cannot give
(its OpenMP merge order is non-deterministic, my_cpp/common.cpp:303-313).

Errors: the reference `printf`s and `exit(1)`s on wrong shapes (collision_manager.cpp:17-26); here a
ValueError is raised instead -- wrong shapes are never silently accepted.
All collision arithmetic runs in the HIP library; there is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int
_c_long = ctypes.c_long
_F16 = ctypes.c_float * 16

_ik_solver = None


def set_ik_solver(fn):
    """Replace the IK feasibility test used when filter_ik=True by a host callback
    fn(ee_in_base: (E,4,4) float32, upper: list[7], lower: list[7]) -> bool array (E,)  -- e.g. a wrapper over ikfast_pybind for
    another robot.  `set_ik_solver(None)` restores the default: the device closed-form solver of the KUKA iiwa14 with the
    redundancy joint fixed at 0 (csrc/iiwa_ik.hip), which is what the reference's generated IKFast file solves
    (my_cpp/common.cpp:9-72)."""
    global _ik_solver
    _ik_solver = fn


def ik_within_limits_device(ee, upper, lower):
    """ee (E,16)/(E,4,4) float32 cuda tensor -> (E,) uint8 cuda tensor: get_ik_within_limits(...).size() > 0 per pose."""
    ee = ee.reshape(-1, 16).contiguous()
    E = ee.shape[0]
    up = (ctypes.c_double * 7)(*[float(v) for v in upper])
    lo = (ctypes.c_double * 7)(*[float(v) for v in lower])
    ok = torch.empty((E,), dtype=torch.uint8, device=ee.device)
    check(L.lib().cg_iiwa_ik_within_limits(_p(ee), ctypes.c_long(E), up, lo, _p(ok), _stream()), 'cg_iiwa_ik_within_limits')
    return ok


def _device():
    if not torch.cuda.is_available():
        raise L.CatgraspAmdError('catgrasp_amd.my_cpp needs a HIP device (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def _mat4(x, name):
    a = np.asarray(x)
    if a.shape != (4, 4):
        raise ValueError(f'{name} shape wrong: {a.shape}')
    return np.ascontiguousarray(a, dtype=np.float32)


def _h16(a):
    return _F16(*a.reshape(-1).tolist())


def _mesh(V, F, what):
    V = np.asarray(V); F = np.asarray(F)
    if V.ndim != 2 or V.shape[1] != 3:
        raise ValueError(f'{what} vertices shape wrong: {V.shape}')
    if F.ndim != 2 or F.shape[1] != 3:
        raise ValueError(f'{what} faces shape wrong: {F.shape}')
    if F.size and (F.min() < 0 or F.max() >= len(V)):
        raise ValueError(f'{what} faces index out of range')
    return np.ascontiguousarray(V, dtype=np.float32), np.ascontiguousarray(F, dtype=np.int32)


def voxelize(pts, resolution, device=None):
    """registerPointCloud (collision_manager.cpp:55-79): occupied octomap leaves of `pts` at `resolution`
    as a device (n,4) int16 tensor (key-32768 per axis), unique and sorted."""
    device = device or _device()
    if isinstance(pts, torch.Tensor):
        t = pts.to(device=device, dtype=torch.float32).contiguous()
    else:
        a = np.asarray(pts)
        if a.ndim != 2 or a.shape[1] != 3:
            raise ValueError(f'point cloud shape wrong: {a.shape}')
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError(f'point cloud shape wrong: {tuple(t.shape)}')
    P = t.shape[0]
    packed = torch.empty((P,), dtype=torch.int64, device=device)
    check(L.lib().cg_voxel_keys(_p(t), _c_long(P), ctypes.c_float(resolution), _p(packed), _stream()), 'cg_voxel_keys')
    uniq = torch.unique(packed)            # sorted
    uniq = uniq[uniq >= 0].contiguous()
    n = uniq.shape[0]
    keys = torch.empty((n, 4), dtype=torch.int16, device=device)
    check(L.lib().cg_unpack_voxel_keys(_p(uniq), _c_long(n), _p(keys), _stream()), 'cg_unpack_voxel_keys')
    return keys


def voxel_blocks(keys):
    """keys (n,4) int16 (voxelize) -> (the keys sorted by Morton code, (ceil(n/64), 2, 4) int16 lowest / highest key of every run of 64):
    with a space-filling order a run of 64 consecutive voxels is a compact blob of space."""
    n = keys.shape[0]
    if n == 0:
        return keys, torch.zeros((0, 2, 4), dtype=torch.int16, device=keys.device)

    def spread(x):                     # 16 bits -> every third bit of 48
        x = x & 0xFFFF
        x = (x | (x << 32)) & 0x1F00000000FFFF
        x = (x | (x << 16)) & 0x1F0000FF0000FF
        x = (x | (x << 8)) & 0x100F00F00F00F00F
        x = (x | (x << 4)) & 0x10C30C30C30C30C3
        x = (x | (x << 2)) & 0x1249249249249249
        return x
    k = keys[:, :3].to(torch.int64) + 32768
    code = (spread(k[:, 0]) << 2) | (spread(k[:, 1]) << 1) | spread(k[:, 2])
    keys = keys[torch.argsort(code)].contiguous()
    nb = (n + 63) // 64
    padded = torch.cat([keys, keys[-1:].expand(nb * 64 - n, 4)]) if nb * 64 != n else keys
    runs = padded.view(nb, 64, 4)
    blocks = torch.stack([runs.min(dim=1).values, runs.max(dim=1).values], dim=1).contiguous()
    return keys, blocks


class CollisionManager:
    """my_cpp.CollisionManager (collision_manager.h:55-69).  Objects are triangle meshes and voxelised point clouds (octomap leaves at
    the registered resolution), each posed by setTransform.  isAnyCollision tests EVERY pair of registered objects like the reference's
    double loop (collision_manager.cpp:93-111):
      {mesh, cloud}  -- the pair the grasp filter forms (common.cpp:176-182): the mesh expressed in the cloud's frame, inv(cloud pose) .
                        mesh pose (the reference never moves its octrees; a posed cloud is the same leaf boxes seen from another frame);
      {mesh, mesh}   -- triangle against triangle, both posed (cg_mesh_mesh_collide);
      {cloud, cloud} -- leaf cube against leaf cube, B's cubes carried into A's frame (cg_voxels_voxels_collide)."""

    def __init__(self):
        self._obs = []
        self._dev = _device()

    def registerMesh(self, V, F):
        V, F = _mesh(V, F, 'mesh')
        self._obs.append({'kind': 'mesh', 'V': torch.from_numpy(V).to(self._dev), 'F': torch.from_numpy(F).to(self._dev),
                          'pose': np.eye(4, dtype=np.float32)})
        return len(self._obs) - 1

    def registerPointCloud(self, pts, resolution):
        keys = voxelize(pts, float(resolution), self._dev)
        self._obs.append({'kind': 'cloud', 'keys': keys, 'res': float(np.float32(resolution)), 'pose': np.eye(4, dtype=np.float32)})
        return len(self._obs) - 1

    def setTransform(self, pose, ob_id):
        pose = _mat4(pose, 'pose')
        ob = self._obs[ob_id]
        if ob['kind'] == 'cloud':
            R = pose[:3, :3].astype(np.float64)
            if not (np.allclose(R.T @ R, np.eye(3), atol=1e-4) and np.linalg.det(R) > 0):
                raise ValueError('a registered point cloud can only be posed rigidly (its leaf boxes are re-expressed, not re-sampled): '
                                 'the rotation block must be orthonormal with determinant +1')
        ob['pose'] = pose

    def _dev_pose(self, m):
        return torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32).reshape(1, 16)).to(self._dev)

    def _pair_collides(self, a, b):
        out = torch.zeros((1,), dtype=torch.uint8, device=self._dev)
        # the pose tensors stay referenced until the launch is queued: a temporary's block would be handed to the next allocation
        if a['kind'] == 'mesh' and b['kind'] == 'mesh':
            pa, pb = self._dev_pose(a['pose']), self._dev_pose(b['pose'])
            check(L.lib().cg_mesh_mesh_collide(_p(a['V']), _p(a['F']), _c_int(a['F'].shape[0]), _p(b['V']), _p(b['F']), _c_int(b['F'].shape[0]),
                                               _p(pa), _p(pb), _p(out), _stream()), 'cg_mesh_mesh_collide')
        elif a['kind'] == 'cloud' and b['kind'] == 'cloud':
            rel = self._dev_pose((np.linalg.inv(a['pose'].astype(np.float64)) @ b['pose'].astype(np.float64)).astype(np.float32))
            check(L.lib().cg_voxels_voxels_collide(_p(a['keys']), _c_int(a['keys'].shape[0]), ctypes.c_float(a['res']), _p(b['keys']),
                                                   _c_int(b['keys'].shape[0]), ctypes.c_float(b['res']), _p(rel), _p(out), _stream()),
                  'cg_voxels_voxels_collide')
        else:
            mesh, cloud = (a, b) if a['kind'] == 'mesh' else (b, a)
            rel = mesh['pose']
            if not np.array_equal(cloud['pose'], np.eye(4, dtype=np.float32)):        # the mesh as seen from the cloud's frame
                rel = (np.linalg.inv(cloud['pose'].astype(np.float64)) @ mesh['pose'].astype(np.float64)).astype(np.float32)
            rel = self._dev_pose(rel)
            check(L.lib().cg_mesh_voxels_collide(_p(mesh['V']), _p(mesh['F']), _c_int(mesh['F'].shape[0]), _p(rel), _c_long(1),
                                                 _p(cloud['keys']), _c_int(cloud['keys'].shape[0]), ctypes.c_float(cloud['res']),
                                                 _p(out), _stream()), 'cg_mesh_voxels_collide')
        return bool(out.item())

    def isAnyCollision(self):
        for i in range(len(self._obs)):
            for j in range(i + 1, len(self._obs)):
                if self._pair_collides(self._obs[i], self._obs[j]):
                    return True
        return False


class _MeshGridC(ctypes.Structure):
    """cg_mesh_grid (include/catgrasp_amd.h)."""
    _fields_ = [('origin', ctypes.c_float * 3), ('cell', ctypes.c_float), ('dims', ctypes.c_int * 3),
                ('cell_start', ctypes.c_void_p), ('tri_ids', ctypes.c_void_p), ('resolution', ctypes.c_float), ('tri_verts', ctypes.c_void_p),
                ('coarse_occupancy', ctypes.c_void_p)]


class MeshGrid:
    """Broad phase for filterGraspPose on large gripper meshes: a uniform grid in the mesh frame whose cells list the
    triangles that can touch a voxel centred in the cell (see cg_mesh_grid).  Purely an accelerator: the narrow phase and
    therefore every result is unchanged.  Built on the device (cg_mesh_grid_count / _fill: one thread per triangle, atomic
    cell counts, prefix sum, per-cell sort); `builder='host'` keeps the numpy construction the device build is tested against."""

    def __init__(self, V, F, resolution, device, cell=None, builder='device', V_dev=None, F_dev=None):
        V = np.asarray(V, dtype=np.float64); F = np.asarray(F, dtype=np.int64)
        res = float(np.float32(resolution))
        inflate = 2.0 * (res * np.sqrt(3.0) / 2.0) + 3e-5          # sigma_min >= 0.5, plus float32 slack
        if cell is None:      # finely tessellated meshes get finer cells: half the candidate pairs per voxel for 8x the (small) cell table
            cell = float(os.environ.get('CATGRASP_AMD_GRID_CELL', 0)) or (max(2.0 * res, 0.001) if len(F) >= 2000 else max(4.0 * res, 0.002))
        lo = V.min(axis=0) - inflate - 1e-6
        hi = V.max(axis=0) + inflate + 1e-6
        dims = np.maximum(np.ceil((hi - lo) / cell).astype(np.int64), 1)
        ncell = int(dims[0] * dims[1] * dims[2])
        if builder == 'host':
            start, tids = self._host_lists(V, F, lo, dims, cell, inflate, ncell)
            self.cell_start = torch.from_numpy(start).to(device)
            self.tri_ids = torch.from_numpy(tids if len(tids) else np.zeros((1,), np.int32)).to(device)
            self.n_entries = int(len(tids))
        else:
            Vd = V_dev if V_dev is not None else torch.from_numpy(np.ascontiguousarray(V, dtype=np.float32)).to(device)
            Fd = F_dev if F_dev is not None else torch.from_numpy(np.ascontiguousarray(F, dtype=np.int32)).to(device)
            org = (ctypes.c_double * 3)(*[float(v) for v in lo])
            dm = (ctypes.c_int * 3)(*[int(v) for v in dims])
            counts = torch.zeros((ncell,), dtype=torch.int32, device=device)
            check(L.lib().cg_mesh_grid_count(_p(Vd), _p(Fd), _c_int(len(F)), org, ctypes.c_double(cell), ctypes.c_double(inflate), dm,
                                             _p(counts), _stream()), 'cg_mesh_grid_count')
            start = torch.zeros((ncell + 1,), dtype=torch.int32, device=device)
            torch.cumsum(counts, 0, out=start[1:])
            self.n_entries = int(start[-1].item())
            self.cell_start = start
            self.tri_ids = torch.empty((max(self.n_entries, 1),), dtype=torch.int32, device=device)
            counts.zero_()
            check(L.lib().cg_mesh_grid_fill(_p(Vd), _p(Fd), _c_int(len(F)), org, ctypes.c_double(cell), ctypes.c_double(inflate), dm,
                                            _p(start), _p(counts), _p(self.tri_ids), _stream()), 'cg_mesh_grid_fill')
        # the triangles as a flat (nf,12) array [v0 v1 v2 pad]: the narrow phase fetches a triangle with three 16-byte loads
        Vf = V_dev if V_dev is not None else torch.from_numpy(np.ascontiguousarray(V, dtype=np.float32)).to(device)
        Ff = F_dev if F_dev is not None else torch.from_numpy(np.ascontiguousarray(F, dtype=np.int32)).to(device)
        self.tri_verts = torch.zeros((max(len(F), 1), 12), dtype=torch.float32, device=device)
        if len(F):
            self.tri_verts[:, :9] = Vf[Ff.long()].reshape(len(F), 9)
        # coarse occupancy: one byte per 4 x 4 x 4 block of cells, 1 iff some cell of it has a triangle list (voxel-block culling)
        nx, ny, nz = (int(v) for v in dims)
        nonempty = (self.cell_start[1:] != self.cell_start[:-1]).view(nx, ny, nz)
        pad = [(-nx) % 4, (-ny) % 4, (-nz) % 4]
        nonempty = torch.nn.functional.pad(nonempty, (0, pad[2], 0, pad[1], 0, pad[0]))
        sx, sy, sz = (nx + 3) // 4, (ny + 3) // 4, (nz + 3) // 4
        self.coarse = nonempty.view(sx, 4, sy, 4, sz, 4).permute(0, 2, 4, 1, 3, 5).reshape(sx, sy, sz, 64).any(dim=3).to(torch.uint8).contiguous()
        self.c = _MeshGridC()
        self.c.tri_verts = self.tri_verts.data_ptr()
        self.c.coarse_occupancy = self.coarse.data_ptr()
        for i in range(3):
            self.c.origin[i] = float(lo[i]); self.c.dims[i] = int(dims[i])
        self.c.cell = float(cell); self.c.resolution = res
        self.c.cell_start = self.cell_start.data_ptr(); self.c.tri_ids = self.tri_ids.data_ptr()

    @staticmethod
    def _host_lists(V, F, lo, dims, cell, inflate, ncell):
        tri = V[F]                                                   # (nf,3,3)
        tlo = np.clip(np.floor((tri.min(axis=1) - inflate - lo) / cell).astype(np.int64), 0, dims - 1)
        thi = np.clip(np.floor((tri.max(axis=1) + inflate - lo) / cell).astype(np.int64), 0, dims - 1)
        cells, tids = [], []
        for t in range(len(F)):
            ii, jj, kk = np.meshgrid(np.arange(tlo[t, 0], thi[t, 0] + 1), np.arange(tlo[t, 1], thi[t, 1] + 1),
                                     np.arange(tlo[t, 2], thi[t, 2] + 1), indexing='ij')
            c = ((ii * dims[1] + jj) * dims[2] + kk).reshape(-1)
            cells.append(c); tids.append(np.full(len(c), t, dtype=np.int32))
        cells = np.concatenate(cells) if cells else np.zeros((0,), dtype=np.int64)
        tids = np.concatenate(tids) if tids else np.zeros((0,), dtype=np.int32)
        order = np.argsort(cells, kind='stable')
        counts = np.bincount(cells, minlength=ncell)
        start = np.zeros(ncell + 1, dtype=np.int32); start[1:] = np.cumsum(counts)
        return start, np.ascontiguousarray(tids[order], dtype=np.int32)


def _digest(*arrays):
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype, a.shape)).encode()); h.update(a.view(np.uint8).reshape(-1).data)
    return h.digest()


class _LRU:
    def __init__(self, cap):
        from collections import OrderedDict
        self.cap, self.d = cap, OrderedDict()

    def get(self, key, make):
        if key in self.d:
            self.d.move_to_end(key)
            return self.d[key]
        v = make()
        self.d[key] = v
        while len(self.d) > self.cap:
            self.d.popitem(last=False)
        return v

    def clear(self):
        self.d.clear()


# The reference rebuilds its BVHs and octrees inside every filterGraspPose call, once per OpenMP thread (common.cpp:176-182).
# Here the device-resident pieces are keyed on the CONTENT of the arrays the 20-argument call passes (a 16-byte blake2b of
# the bytes: ~0.1 ms for a 20k-point cloud), so the per-object calls of grasp_sampler.py:216 and :345, which pass the same
# gripper meshes every time and the same clouds twice per object, reuse them.  clear_scene_cache() drops everything.
_mesh_cache = _LRU(8)       # (mesh bytes, resolution, device) -> (V, F device tensors, MeshGrid | None)
_cloud_cache = _LRU(32)     # (cloud bytes, resolution, device) -> voxel keys


def clear_scene_cache():
    _mesh_cache.clear(); _cloud_cache.clear()


class GripperScene:
    """Device-resident inputs of filterGraspPose that do not change between calls on one object:
    gripper meshes and the two voxelised collision clouds.  Build once, filter many pose batches."""

    def __init__(self, gripper_vertices, gripper_faces, gripper_enclosed_vertices, gripper_enclosed_faces,
                 gripper_collision_pts, gripper_enclosed_collision_pts, octo_resolution, device=None, accel=True, cache=True):
        dev = device or _device()
        self.device = dev
        self.res = float(np.float32(octo_resolution))

        def mesh(V, F, what):
            V, F = _mesh(V, F, what)

            def make():
                Vd = torch.from_numpy(V).to(dev); Fd = torch.from_numpy(F).to(dev)
                grid = MeshGrid(V, F, self.res, dev, V_dev=Vd, F_dev=Fd) if accel and len(F) else None
                return Vd, Fd, grid
            return _mesh_cache.get((_digest(V, F), self.res, str(dev), bool(accel)), make) if cache else make()

        def cloud(pts):
            # -> (voxel keys in Morton order, key box of every run of 64): see voxel_blocks
            if isinstance(pts, torch.Tensor) or not cache:
                return voxel_blocks(voxelize(pts, self.res, dev))
            a = np.asarray(pts)
            if a.ndim != 2 or a.shape[1] != 3:
                raise ValueError(f'point cloud shape wrong: {a.shape}')
            a = np.ascontiguousarray(a, dtype=np.float32)
            return _cloud_cache.get((_digest(a), self.res, str(dev)), lambda: voxel_blocks(voxelize(a, self.res, dev)))

        self.V, self.F, self.grid_open = mesh(gripper_vertices, gripper_faces, 'gripper')
        self.Ve, self.Fe, self.grid_enc = mesh(gripper_enclosed_vertices, gripper_enclosed_faces, 'gripper_enclosed')
        # the voxel sets in Morton order + the key box of every run of 64 (voxel_blocks): the grid kernel skips runs that cannot touch a
        # triangle list.  Which voxels collide does not depend on their order.
        self.keys_open, self.blocks_open = cloud(gripper_collision_pts)
        self.keys_bg, self.blocks_bg = cloud(gripper_enclosed_collision_pts)


class _FilterSegmentC(ctypes.Structure):
    """cg_filter_segment (include/catgrasp_amd.h)."""
    _fields_ = [('grasp_poses', ctypes.c_void_p), ('symmetry_tfs', ctypes.c_void_p), ('n_pose', ctypes.c_int), ('n_sym', ctypes.c_int),
                ('nocs_pose', ctypes.c_float * 16), ('canonical_to_nocs', ctypes.c_float * 16), ('c2c', ctypes.c_float * 16),
                ('adjust_collision_pose', ctypes.c_int), ('n_open_keys', ctypes.c_int),
                ('open_keys', ctypes.c_void_p), ('open_blocks', ctypes.c_void_p), ('bg_keys', ctypes.c_void_p), ('bg_blocks', ctypes.c_void_p),
                ('n_bg_keys', ctypes.c_int), ('reserved', ctypes.c_int), ('first', ctypes.c_longlong)]


class FilterPlan:
    """Several filterGraspPose calls prepared as ONE launch sequence (cg_filter_grasp_pose_multi): the segment table on the host and its
    device copy.  segments: [(scene, grasp_poses (n,16) f32 cuda, symmetry_tfs (m,16) f32 cuda, nocs_pose 4x4, canonical_to_nocs 4x4,
    adjust_collision_pose)], all scenes on one device with the SAME gripper meshes and resolution (one gripper per run, as in the
    reference).  The plan keeps references to the tensors its table points at; build it once for a fixed set of calls (bench step,
    pick cycle) and run it as often as needed -- run() itself moves no table and synchronises nothing."""

    def __init__(self, segments):
        if not segments:
            raise ValueError('FilterPlan needs at least one segment')
        self.scene0 = sc0 = segments[0][0]
        self.keep = []
        tab = (_FilterSegmentC * len(segments))()
        for r, (scene, gp, st, nocs_pose, c2n, adjust) in zip(tab, segments):
            if scene.device != sc0.device or scene.res != sc0.res or scene.V is not sc0.V or scene.F is not sc0.F or scene.Ve is not sc0.Ve \
                    or scene.Fe is not sc0.Fe:
                raise ValueError('the segments of one FilterPlan must share the gripper meshes, the resolution and the device')
            for t, name in ((gp, 'grasp_poses'), (st, 'symmetry_tfs')):
                if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 16 and t.is_contiguous()):
                    raise ValueError(f'{name}: a contiguous (n,16) float32 device tensor is expected')
            self.keep.append((scene, gp, st))
            r.grasp_poses, r.symmetry_tfs, r.n_pose, r.n_sym = gp.data_ptr(), st.data_ptr(), gp.shape[0], st.shape[0]
            r.nocs_pose[:] = [float(v) for v in _mat4(nocs_pose, 'nocs_pose').reshape(16)]
            r.canonical_to_nocs[:] = [float(v) for v in _mat4(c2n, 'canonical_to_nocs_transform').reshape(16)]
            r.adjust_collision_pose = int(bool(adjust))
            r.open_keys, r.n_open_keys = scene.keys_open.data_ptr(), scene.keys_open.shape[0]
            r.bg_keys, r.n_bg_keys = scene.keys_bg.data_ptr(), scene.keys_bg.shape[0]
            bo, bb = getattr(scene, 'blocks_open', None), getattr(scene, 'blocks_bg', None)
            r.open_blocks = bo.data_ptr() if bo is not None and r.n_open_keys else None
            r.bg_blocks = bb.data_ptr() if bb is not None and r.n_bg_keys else None
        L.lib().cg_filter_segments_prepare.restype = ctypes.c_long
        E = L.lib().cg_filter_segments_prepare(tab, _c_int(len(segments)))
        if E < 0:
            raise L.CatgraspAmdError(f'cg_filter_segments_prepare failed with status {E}')
        self.E, self.table, self.n = int(E), tab, len(segments)
        self.firsts = [int(r.first) for r in tab]
        self.counts = [int(r.n_pose) * int(r.n_sym) for r in tab]
        # one upload per plan, from page-locked memory so that it is queued behind the stream's work instead of waiting for it
        self._host = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).pin_memory()
        self.d_table = self._host.to(sc0.device, non_blocking=True)

    def run(self, gripper_in_grasp, filter_approach_dir_face_camera, keep_rejected_pose=False, work_stats=None, ik_ok=None):
        """-> codes (E) int8, poses (E,4,4) float32, nudge (E) int8 over the plan's evaluations, segment after segment."""
        sc, dev = self.scene0, self.scene0.device
        codes = torch.empty((self.E,), dtype=torch.int8, device=dev)
        poses = torch.empty((self.E, 16), dtype=torch.float32, device=dev)
        nudge = torch.empty((self.E,), dtype=torch.int8, device=dev)
        if self.E == 0:
            return codes, poses.view(0, 4, 4), nudge
        go = ctypes.byref(sc.grid_open.c) if sc.grid_open is not None else None
        ge = ctypes.byref(sc.grid_enc.c) if sc.grid_enc is not None else None
        check(L.lib().cg_filter_grasp_pose_multi(self.table, _p(self.d_table), _c_int(self.n), _h16(_mat4(gripper_in_grasp, 'gripper_in_grasp')),
                                                 _c_int(int(bool(filter_approach_dir_face_camera))), _p(ik_ok),
                                                 _p(sc.V), _p(sc.F), _c_int(sc.F.shape[0]), _p(sc.Ve), _p(sc.Fe), _c_int(sc.Fe.shape[0]),
                                                 ctypes.c_float(sc.res), _p(codes), _p(poses), _p(nudge), go, ge, _c_int(int(bool(keep_rejected_pose))),
                                                 _p(work_stats), _stream()), 'cg_filter_grasp_pose_multi')
        return codes, poses.view(self.E, 4, 4), nudge


def filter_on_device(scene, grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs_transform, cam_in_world, ee_in_grasp,
                     gripper_in_grasp, filter_approach_dir_face_camera, filter_ik, adjust_collision_pose, upper=None, lower=None,
                     keep_rejected_pose=False, work_stats=None):
    """Device-tensor form: grasp_poses (n,4,4)/(n,16) and symmetry_tfs (m,4,4) float32 cuda tensors (or arrays).
    Returns codes (E) int8, poses (E,4,4) float32, nudge (E) int8 as cuda tensors, E = n*m in input order.
    keep_rejected_pose: rejected evaluations keep their composed grasp_in_cam in `poses` instead of zeros.
    work_stats: optional (3,) int64 cuda tensor the grid kernel adds its memory work to (voxel keys read, grid cells looked up,
    (voxel, triangle) pairs tested) -- measurement only."""
    dev = scene.device

    def dev_poses(x, name):
        if isinstance(x, torch.Tensor):
            t = x.to(device=dev, dtype=torch.float32)
        else:
            a = np.asarray(x, dtype=np.float64 if len(x) else np.float32)
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        if t.numel() == 0:
            return t.reshape(0, 16)
        if t.shape[-2:] != (4, 4) and t.shape[-1] != 16:
            raise ValueError(f'{name} shape wrong: {tuple(t.shape)}')
        return t.reshape(-1, 16).contiguous()

    gp = dev_poses(grasp_poses, 'grasp_poses')
    st = dev_poses(symmetry_tfs, 'symmetry_tfs')
    n_pose, n_sym = gp.shape[0], st.shape[0]
    E = n_pose * n_sym
    mats = [_mat4(m, nm) for m, nm in ((nocs_pose, 'nocs_pose'), (canonical_to_nocs_transform, 'canonical_to_nocs_transform'),
                                       (cam_in_world, 'cam_in_world'), (ee_in_grasp, 'ee_in_grasp'), (gripper_in_grasp, 'gripper_in_grasp'))]
    hm = [_h16(m) for m in mats]
    codes = torch.empty((E,), dtype=torch.int8, device=dev)
    poses = torch.empty((E, 16), dtype=torch.float32, device=dev)
    nudge = torch.empty((E,), dtype=torch.int8, device=dev)

    def launch(ik_ok, ee_out):
        go = ctypes.byref(scene.grid_open.c) if scene.grid_open is not None else None
        ge = ctypes.byref(scene.grid_enc.c) if scene.grid_enc is not None else None
        check(L.lib().cg_filter_grasp_pose_accel(
            _p(gp), _c_int(n_pose), _p(st), _c_int(n_sym), hm[0], hm[1], hm[2], hm[3], hm[4],
            _c_int(int(bool(filter_approach_dir_face_camera))), _c_int(int(bool(adjust_collision_pose))), _p(ik_ok),
            _p(scene.V), _p(scene.F), _c_int(scene.F.shape[0]), _p(scene.Ve), _p(scene.Fe), _c_int(scene.Fe.shape[0]),
            _p(scene.keys_open), _c_int(scene.keys_open.shape[0]), _p(scene.keys_bg), _c_int(scene.keys_bg.shape[0]),
            ctypes.c_float(scene.res), _p(codes), _p(poses), _p(nudge), _p(ee_out), go, ge, _c_int(int(bool(keep_rejected_pose))), _p(work_stats),
            _p(getattr(scene, 'blocks_open', None)), _p(getattr(scene, 'blocks_bg', None)), _stream()),
              'cg_filter_grasp_pose_accel')

    ik_ok = None
    if filter_ik and E > 0:
        if upper is None or lower is None or len(upper) != 7 or len(lower) != 7:
            raise ValueError('filter_ik=True needs the 7 upper and 7 lower joint limits')
        ee = torch.empty((E, 16), dtype=torch.float32, device=dev)
        launch(None, ee)
        if _ik_solver is None:       # default: iiwa14 closed form on the device, no host round trip
            ik_ok = ik_within_limits_device(ee, upper, lower)
        else:
            ok = np.asarray(_ik_solver(ee.cpu().numpy().reshape(E, 4, 4), list(upper), list(lower))).astype(np.uint8).reshape(E)
            ik_ok = torch.from_numpy(ok).to(dev)
    if E > 0:
        launch(ik_ok, None)
    return codes, poses.view(E, 4, 4), nudge


def filterGraspPoseDetailed(grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs_transform, cam_in_world, ee_in_grasp,
                            gripper_in_grasp, filter_approach_dir_face_camera, filter_ik, adjust_collision_pose, upper, lower,
                            gripper_vertices, gripper_faces, gripper_enclosed_vertices, gripper_enclosed_faces,
                            gripper_collision_pts, gripper_enclosed_collision_pts, octo_resolution, verbose=False, accel=True):
    """filterGraspPose in input order: returns (codes (E,) int8, poses (E,4,4) float32, nudge (E,) int8) numpy,
    E = len(grasp_poses)*len(symmetry_tfs), e = i*len(symmetry_tfs)+j.
    codes: 0 keep, 1 approach-dir, 2 IK, 3 open-gripper collision / no nudge found, 4 enclosed-gripper collision."""
    scene = GripperScene(gripper_vertices, gripper_faces, gripper_enclosed_vertices, gripper_enclosed_faces,
                         gripper_collision_pts, gripper_enclosed_collision_pts, octo_resolution, accel=accel)
    codes, poses, nudge = filter_on_device(scene, grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs_transform, cam_in_world,
                                           ee_in_grasp, gripper_in_grasp, filter_approach_dir_face_camera, filter_ik,
                                           adjust_collision_pose, upper, lower)
    codes = codes.cpu().numpy(); poses = poses.cpu().numpy(); nudge = nudge.cpu().numpy()
    if verbose:
        # counters of common.cpp:316-319
        print('n_approach_dir_rej=%d, n_ik_rej=%d, n_open_gripper_rej=%d, n_close_gripper_rej=%d' % (
            int((codes == 1).sum()), int((codes == 2).sum()), int((codes == 3).sum()), int((codes == 4).sum())))
    return codes, poses, nudge


def filterGraspPose(grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs_transform, cam_in_world, ee_in_grasp,
                    gripper_in_grasp, filter_approach_dir_face_camera, filter_ik, adjust_collision_pose, upper, lower,
                    gripper_vertices, gripper_faces, gripper_enclosed_vertices, gripper_enclosed_faces,
                    gripper_collision_pts, gripper_enclosed_collision_pts, octo_resolution, verbose):
    """my_cpp.filterGraspPose (common.cpp:156-321): list of surviving grasp_in_cam 4x4 float32 matrices
    (here in deterministic input order; the reference's order is thread-dependent)."""
    codes, poses, _ = filterGraspPoseDetailed(grasp_poses, symmetry_tfs, nocs_pose, canonical_to_nocs_transform, cam_in_world,
                                              ee_in_grasp, gripper_in_grasp, filter_approach_dir_face_camera, filter_ik,
                                              adjust_collision_pose, upper, lower, gripper_vertices, gripper_faces,
                                              gripper_enclosed_vertices, gripper_enclosed_faces, gripper_collision_pts,
                                              gripper_enclosed_collision_pts, octo_resolution, verbose)
    return list(poses[codes == 0])        # one fancy-index copy; the rows are views into it (the reference returns fresh arrays too)


def _float_loop_count(limit, step):
    """Trip count of `for (float v=0; v<limit; v+=step)` with float32 accumulation (common.cpp:123,144)."""
    step = np.float32(step); limit = np.float32(limit)
    if not step > 0:
        raise ValueError('step must be positive')
    v = np.float32(0); n = 0
    while v < limit:
        v = np.float32(v + step); n += 1
    return n


def augmentGraspPoses(R0, selected_point, sphere_pts, inplane_rot_step, hand_depth, approach_step, init_bite):
    """my_cpp.augmentGraspPoses (common.cpp:118-153) -> list of 4x4 float32 poses.
    Note: the reference loops `i < sphere_pts.size()` (= 3*rows, common.cpp:119) and reads rows past the end of the
    matrix; only the valid rows are used here."""
    dev = _device()
    R0 = np.ascontiguousarray(np.asarray(R0, dtype=np.float32))
    if R0.shape != (3, 3):
        raise ValueError(f'R0 shape wrong: {R0.shape}')
    p = np.ascontiguousarray(np.asarray(selected_point, dtype=np.float32).reshape(-1))
    if p.shape != (3,):
        raise ValueError(f'selected_point shape wrong: {p.shape}')
    sp = np.asarray(sphere_pts, dtype=np.float32)
    if sp.size and (sp.ndim != 2 or sp.shape[1] != 3):
        raise ValueError(f'sphere_pts shape wrong: {sp.shape}')
    sp = np.ascontiguousarray(sp.reshape(-1, 3))
    n_rot = _float_loop_count(180.0, inplane_rot_step)
    n_depth = _float_loop_count(hand_depth, approach_step)
    total = (1 + len(sp) * n_rot) * n_depth
    out = torch.empty((total, 16), dtype=torch.float32, device=dev)
    sp_d = torch.from_numpy(sp).to(dev)
    F9 = (ctypes.c_float * 9)(*R0.reshape(-1).tolist())
    F3 = (ctypes.c_float * 3)(*p.tolist())
    check(L.lib().cg_augment_grasp_poses(F9, F3, _p(sp_d), _c_int(len(sp)), _c_int(n_rot), ctypes.c_float(inplane_rot_step), _c_int(n_depth),
                                         ctypes.c_float(approach_step), ctypes.c_float(init_bite), _p(out), _stream()),
          'cg_augment_grasp_poses')
    poses = out.cpu().numpy().reshape(total, 4, 4)
    return list(poses)


def makeOccupancyGridFromCloudScan(pts, K, resolution, return_tensor=False):
    """my_cpp.makeOccupancyGridFromCloudScan (common.cpp:324-431): every point of the 5 mm-padded lattice around the
    scan that lies at or behind the observed surface along its camera ray -> (Q,3) float32.
    `K` is accepted for signature compatibility (the reference only uses it for unused pixel bounds, :336-347).
    Output is in lattice order (x slowest); the reference's order is thread-dependent."""
    dev = _device()
    a = np.asarray(pts)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f'point cloud shape wrong: {a.shape}')
    a = np.ascontiguousarray(a, dtype=np.float32)
    if len(a) == 0:
        return np.zeros((0, 3), dtype=np.float32)
    res = np.float32(resolution)
    keys = voxelize(a, float(res), dev)
    if keys.shape[0] == 0:
        return np.zeros((0, 3), dtype=np.float32)
    kmin = keys[:, :3].min(dim=0).values.cpu().numpy().astype(np.int64)
    kmax = keys[:, :3].max(dim=0).values.cpu().numpy().astype(np.int64)
    dims = (kmax - kmin + 1)
    nbits = int(dims[0]) * int(dims[1]) * int(dims[2])
    if nbits > (1 << 33):
        raise MemoryError(f'occupancy bitmap of {nbits} bits is too large; use a coarser resolution')
    bits = torch.zeros(((nbits + 31) // 32,), dtype=torch.int32, device=dev)
    check(L.lib().cg_occupancy_set_bits(_p(keys), _c_long(keys.shape[0]), _p(bits), _c_int(int(kmin[0])), _c_int(int(kmin[1])),
                                        _c_int(int(kmin[2])), _c_int(int(dims[0])), _c_int(int(dims[1])), _c_int(int(dims[2])), _stream()),
          'cg_occupancy_set_bits')
    # lattice bounds exactly as common.cpp:353-376 (float32 arithmetic)
    mx = a.max(axis=0); mn = a.min(axis=0)
    pad = np.float32(0.005)
    n = [int(np.float32(np.float32(np.float32(mx[i] + pad) - np.float32(mn[i] - pad)) / res)) for i in range(3)]
    origin = [np.float32(mn[i] - pad) for i in range(3)]
    max_range = float(np.float32(np.sqrt(float(np.float32(mx[0] + pad)) ** 2 + float(np.float32(mx[1] + pad)) ** 2 +
                                         float(np.float32(mx[2] + pad)) ** 2)))
    total = max(n[0], 0) * max(n[1], 0) * max(n[2], 0)
    if total == 0:
        return np.zeros((0, 3), dtype=np.float32)
    lattice = torch.empty((total, 3), dtype=torch.float32, device=dev)
    keep = torch.empty((total,), dtype=torch.uint8, device=dev)
    check(L.lib().cg_occupancy_grid_rays(_p(bits), _c_int(int(kmin[0])), _c_int(int(kmin[1])), _c_int(int(kmin[2])), _c_int(int(dims[0])),
                                         _c_int(int(dims[1])), _c_int(int(dims[2])), ctypes.c_float(origin[0]), ctypes.c_float(origin[1]),
                                         ctypes.c_float(origin[2]), ctypes.c_float(res), _c_int(n[0]), _c_int(n[1]), _c_int(n[2]),
                                         ctypes.c_double(max_range), _p(lattice), _p(keep), _stream()), 'cg_occupancy_grid_rays')
    out = lattice[keep.bool()]
    return out if return_tensor else out.cpu().numpy()
