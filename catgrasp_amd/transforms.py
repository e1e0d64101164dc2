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


class NumpyChoiceStream:
    """numpy's GLOBAL generator replayed in C (cg_host_numpy_choice_rows, csrc/nprng.hip): hands out, chunk by chunk, exactly the
    rows `np.random.choice(np.arange(n_valid), n_pts, replace=n_valid<n_pts)` would return call after call, and puts the advanced
    Mersenne-Twister state back into numpy on close().  The C call holds no GIL, so a worker thread can draw the next chunk while
    the device scores the current one."""

    def __init__(self, n_valid, n_pts, state=None):
        """state: an explicit generator state (get_state() layout) to replay from instead of numpy's current one -- drawing AHEAD of
        the point the global generator has reached (pipeline.evaluate_objects); close() then is the caller's decision."""
        import ctypes
        from . import _lib as L
        self._ct, self._fn = ctypes, L.lib().cg_host_numpy_choice_rows
        self.n_valid, self.n_pts = int(n_valid), int(n_pts)
        st = np.random.get_state() if state is None else state
        if st[0] != 'MT19937':
            raise RuntimeError(f'numpy global generator is {st[0]}, expected the legacy MT19937')
        self._rest = (st[3], st[4])
        self._key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        self._pos = ctypes.c_int(int(st[2]))
        self._scratch = np.empty((max(self.n_valid, 1),), dtype=np.int32)

    def state(self):
        """The generator state this replay has reached (get_state() layout, own copy of the key)."""
        return ('MT19937', self._key.copy(), int(self._pos.value)) + self._rest

    def draw(self, count, out=None):
        ct = self._ct
        if out is None:
            out = np.empty((count, self.n_pts), dtype=np.int32)
        assert out.shape == (count, self.n_pts) and out.dtype == np.int32 and out.flags.c_contiguous
        if self.on_device_chain:         # replace=False, n_valid <= 65536: vectorised partner extraction + the swap chain in L1
            from . import _lib as L
            rc = L.lib().cg_host_numpy_permutation_rows(self._key.ctypes.data_as(ct.c_void_p), ct.byref(self._pos), ct.c_int(self.n_valid),
                                                        ct.c_int(self.n_pts), ct.c_long(count), ct.c_int(0), out.ctypes.data_as(ct.c_void_p))
            if rc != 0:
                raise RuntimeError(f'cg_host_numpy_permutation_rows failed with status {rc}')
            return out
        rc = self._fn(self._key.ctypes.data_as(ct.c_void_p), ct.byref(self._pos), ct.c_int(self.n_valid), ct.c_int(self.n_pts),
                      ct.c_long(count), self._scratch.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p))
        if rc != 0:
            raise RuntimeError(f'cg_host_numpy_choice_rows failed with status {rc}')
        return out

    @property
    def on_device_chain(self):
        """True when the swap chain of the draw can run on the device (replace=False rows of a cloud that fits the u16 LDS
        permutation): the host then only extracts the swap partners from the stream (draw_partners)."""
        return self.n_pts <= self.n_valid <= 65536 and self.n_valid >= 2

    @property
    def partner_stride(self):
        return (self.n_valid - 1 + 7) & ~7

    def draw_partners(self, count, out=None):
        """The sequential part of `count` permutation(n_valid) draws alone (cg_host_numpy_shuffle_partners): the Fisher-Yates swap
        partners as a (count, stride) uint16 array, stride = n_valid-1 rounded up to 8; ops.apply_shuffle_rows turns them into the
        rows draw() would have returned.  Advances the generator exactly like draw()."""
        ct = self._ct
        from . import _lib as L
        stride = self.partner_stride
        if out is None:
            out = np.empty((count, stride), dtype=np.uint16)
        assert out.shape == (count, stride) and out.dtype == np.uint16 and out.flags.c_contiguous
        rc = L.lib().cg_host_numpy_shuffle_partners(self._key.ctypes.data_as(ct.c_void_p), ct.byref(self._pos), ct.c_int(self.n_valid),
                                                    ct.c_long(count), ct.c_long(stride), out.ctypes.data_as(ct.c_void_p))
        if rc != 0:
            raise RuntimeError(f'cg_host_numpy_shuffle_partners failed with status {rc}')
        return out

    def close(self):
        np.random.set_state(('MT19937', self._key, int(self._pos.value)) + self._rest)


class NumpyHeadsDraw:
    """`count` consecutive draws of `np.random.choice(n, size=k, replace=False)` from numpy's GLOBAL generator, replayed in C
    (cg_host_numpy_choice_heads, csrc/nprng_heads.hip: only the k heads of each permutation are materialised).  The state is taken
    at construction; with `pool` (a ThreadPoolExecutor) the draw runs there without the GIL while the caller queues device work.
    result() -> (count,k) int32 and puts the advanced state back into numpy; cancel() waits and leaves numpy's state alone."""

    HEAD_MAX, N_MAX = 16, 65536

    def __init__(self, n, k, count, pool=None, isa=0, state=None):
        import ctypes
        from . import _lib as L
        if not (2 <= n <= self.N_MAX and 1 <= k <= min(n, self.HEAD_MAX)):
            raise ValueError(f'NumpyHeadsDraw: n={n}, k={k} outside 2 <= n <= {self.N_MAX}, 1 <= k <= min(n, {self.HEAD_MAX})')
        st = np.random.get_state() if state is None else state          # explicit state: a draw ahead of the global generator
        if st[0] != 'MT19937':
            raise RuntimeError(f'numpy global generator is {st[0]}, expected the legacy MT19937')
        self._rest = (st[3], st[4])
        self._key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        self._pos = ctypes.c_int(int(st[2]))
        self._out = np.empty((count, k), dtype=np.int32)
        fn, ct = L.lib().cg_host_numpy_choice_heads, ctypes

        self.seconds = None                  # duration of the C call (on whichever thread ran it)

        def run():
            import time
            t0 = time.perf_counter()
            rc = fn(self._key.ctypes.data_as(ct.c_void_p), ct.byref(self._pos), ct.c_int(n), ct.c_int(k), ct.c_long(count),
                    ct.c_int(isa), self._out.ctypes.data_as(ct.c_void_p))
            self.seconds = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError(f'cg_host_numpy_choice_heads failed with status {rc}')
            return self._out
        self._future = pool.submit(run) if pool is not None else None
        self._run = run

    def result(self, set_state=True):
        out = self._future.result() if self._future is not None else self._run()
        if set_state:
            np.random.set_state(('MT19937', self._key, int(self._pos.value)) + self._rest)
        return out

    def state(self):
        """The generator state after the draw (valid once result() returned)."""
        return ('MT19937', self._key.copy(), int(self._pos.value)) + self._rest

    def cancel(self):
        if self._future is not None:
            self._future.exception()


def same_state(a, b):
    """Two generator states (get_state() layout) denote the same point of the stream."""
    return a[0] == b[0] and int(a[2]) == int(b[2]) and int(a[3]) == int(b[3]) and float(a[4]) == float(b[4]) and np.array_equal(a[1], b[1])


def advance_choice_rows(state, n_valid, n_pts, count, piece=1024):
    """The generator state after `count` further draws of np.random.choice(np.arange(n_valid), n_pts, replace=n_valid<n_pts) from
    `state`, without keeping the rows: where the stream WILL stand once a scoring pass has drawn its resampling indices."""
    st = NumpyChoiceStream(n_valid, n_pts, state=state)
    if st.on_device_chain:
        buf = np.empty((min(piece, max(count, 1)), st.partner_stride), dtype=np.uint16)
        draw = st.draw_partners
    else:
        buf = np.empty((min(piece, max(count, 1)), n_pts), dtype=np.int32)
        draw = st.draw
    done = 0
    while done < count:
        c = min(len(buf), count - done)
        draw(c, out=buf[:c])
        done += c
    return st.state()


def draw_choice_heads(n, k, count):
    """[np.random.choice(n, size=k, replace=False) for _ in range(count)] from numpy's global generator (same rows, same state
    afterwards).  -> (count,k) int32.  Shapes outside NumpyHeadsDraw's take the full-row replay."""
    if 2 <= n <= NumpyHeadsDraw.N_MAX and 1 <= k <= min(n, NumpyHeadsDraw.HEAD_MAX):
        return NumpyHeadsDraw(n, k, count).result()
    return draw_ids_reference(n, k, count)


def draw_ids_reference(n_valid, n_pts, count):
    """Resample indices drawn exactly as the reference does: one `np.random.choice` per sample from
    numpy's GLOBAL generator (dataset_grasp.py:72-73), with replacement iff n_valid < n_pts.
    Seeding numpy therefore reproduces the reference's draws (the stream is replayed in C: same outputs, same generator state
    afterwards, tests/test_cabi_and_host.py).  -> (count, n_pts) int32."""
    st = NumpyChoiceStream(n_valid, n_pts)
    try:
        return st.draw(count)
    finally:
        st.close()


_draw_counter = [0]


def draw_ids_device(n_valid, n_pts, count, device, generator=None, seed=None, base=0, out=None, row_offset=0):
    """The same draw on the device (cg_draw_resample_ids; NOT numpy's stream): per row a uniform n_pts-subset of
    [0,n_valid) in uniform order when n_valid >= n_pts (= np.random.choice(replace=False)), iid uniform indices otherwise
    (= replace=True).  Counter-based: `seed` (or the next draw of `generator`, or a process-wide counter seeded from numpy's
    global generator) and the global row index `row_offset + r` fix every row, so a shard of a batch draws what the whole batch
    would.  -> (count, n_pts) int32 cuda tensor, each id offset by `base`."""
    import ctypes
    from . import _lib as L
    if seed is None:
        if generator is not None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=generator.device).item())
        else:
            if _draw_counter[0] == 0:
                _draw_counter[0] = int(np.random.randint(1, 2 ** 31)) << 20
            _draw_counter[0] += 1
            seed = _draw_counter[0]
    if out is None:
        out = torch.empty((count, n_pts), dtype=torch.int32, device=device)
    assert out.shape == (count, n_pts) and out.dtype == torch.int32 and out.is_contiguous() and out.is_cuda
    st = L.lib().cg_draw_resample_ids(ctypes.c_int(n_valid), ctypes.c_int(n_pts), ctypes.c_long(count), ctypes.c_ulonglong(seed & (2 ** 64 - 1)),
                                      ctypes.c_int(base), ctypes.c_long(row_offset), L._p(out), L._stream())
    if st == -2:        # CG_ERR_UNSUPPORTED: a shape outside the kernel's (without replacement from > 65535 points; n_pts % 4 != 0)
        gen = torch.Generator(device=device); gen.manual_seed((seed + 0x9E3779B97F4A7C15 * (row_offset + 1)) % (2 ** 63))
        if n_valid < n_pts:
            out.copy_(torch.randint(0, n_valid, (count, n_pts), device=device, generator=gen, dtype=torch.int32) + base)
        else:
            out.copy_(torch.rand((count, n_valid), device=device, generator=gen).argsort(dim=1)[:, :n_pts].to(torch.int32) + base)
        return out
    L.check(st, 'cg_draw_resample_ids')
    return out


def pose_inverse_rows_device(poses, center):
    """pose_inverse_rows for poses that already live on the device: (E,4,4)/(E,16) float32 cuda tensor -> (E,12) float32
    (cg_pose_inverse_rows: float64 arithmetic on the device, rounded once)."""
    import ctypes
    from . import _lib as L
    p = poses.reshape(-1, 16)
    assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
    out = torch.empty((p.shape[0], 12), dtype=torch.float32, device=p.device)
    c = (ctypes.c_double * 3)(*[float(v) for v in np.asarray(center, dtype=np.float64).reshape(3)])
    L.check(L.lib().cg_pose_inverse_rows(L._p(p), ctypes.c_long(p.shape[0]), c, L._p(out), L._stream()), 'cg_pose_inverse_rows')
    return out


class DeviceCloud:
    """An object cloud resident in HBM: z-filtered, centred on its centroid in float64 and rounded once to
    float32 (camera-frame coordinates are ~0.6 m with ~1 cm extent: centring keeps 3 more decimal digits)."""

    def __init__(self, cloud_xyz, cloud_normal, device):
        xyz = np.asarray(cloud_xyz, dtype=np.float64)
        nrm = np.asarray(cloud_normal, dtype=np.float64)
        if xyz.ndim != 2 or xyz.shape[1] != 3 or nrm.shape != xyz.shape:
            raise ValueError(f'cloud_xyz {xyz.shape} / cloud_normal {nrm.shape} must both be (M,3)')
        # the fused MLP kernels max-pool with NaN-ignoring arithmetic (-fno-honor-nans): a NaN/Inf point would be silently
        # pooled away instead of poisoning the result as it does in the reference, so it is rejected at the boundary.  (A NaN or an
        # Inf anywhere makes the sum non-finite -- Inf - Inf = NaN: two reductions in the usual case instead of two element-wise
        # passes; this constructor is a quarter of the host time of a predict_batch call of a few poses.)
        with np.errstate(invalid='ignore', over='ignore'):
            quick = np.isfinite(xyz.sum()) and np.isfinite(nrm.sum())
        if not quick and not (np.isfinite(xyz).all() and np.isfinite(nrm).all()):
            raise ValueError('cloud_xyz / cloud_normal contain NaN or Inf')
        z = xyz[:, 2] if len(xyz) else xyz.reshape(-1)
        if len(xyz) and z.min() >= 0.1:          # the usual case (the rig's clouds sit at z ~ 0.6 m): nothing is filtered, nothing is copied
            self.keep_ids = np.arange(len(xyz))
            self.xyz64, self.normal64 = xyz, nrm
        else:
            m = valid_mask(xyz)
            self.keep_ids = np.arange(len(xyz))[m]
            self.xyz64 = xyz[m].reshape(-1, 3)
            self.normal64 = nrm[m].reshape(-1, 3)
        self.n = len(self.xyz64)
        self.center = self.xyz64.mean(axis=0) if self.n else np.zeros(3)
        n_pad = (self.n + 3) & ~3                                   # ONE upload for coordinates and normals (a pageable copy blocks the
        both = np.empty((2, n_pad, 3), dtype=np.float32)            # caller); rows padded to 4 points so that both halves stay 16-byte aligned
        np.subtract(self.xyz64, self.center, out=both[0, :self.n], casting='same_kind')      # float64 difference, rounded once
        both[1, :self.n] = self.normal64
        both[:, self.n:] = 0
        both = torch.from_numpy(both).to(device)
        self.xyz, self.normal = both[0, :self.n], both[1, :self.n]
        self.device = device


def pose_inverse_rows(grasp_poses, center):
    """inv(grasp_pose) in float64 (dataset_grasp.py:69-70 use np.linalg.inv), re-expressed for a cloud
    shifted by -center, rounded to float32: rows (G,12) of [R | t] with x_grasp = R x_centred + t."""
    P = np.asarray(grasp_poses, dtype=np.float64).reshape(-1, 4, 4)
    if len(P) == 0:
        return np.zeros((0, 12), dtype=np.float32)
    if not np.isfinite(P).all():
        raise ValueError('grasp_poses contain NaN or Inf')
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
