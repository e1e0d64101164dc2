"""Seeded synthetic inputs for tests and bench: the reference ships no weights,
gripper meshes, object models or datasets, so parity and throughput are measured on
synthetic checkpoints / clouds / candidates / gripper of the same shapes.  numpy only."""
import numpy as np


# ---------------------------------------------------------------------------------------------
# reference-layout checkpoints (parameter names of pointnet2.py:153-329)
# ---------------------------------------------------------------------------------------------
def _stn_shapes(p, cin, k):
    s = {}
    for name, (o, i) in {'conv1': (64, cin), 'conv2': (128, 64), 'conv3': (1024, 128)}.items():
        s[p + name + '.weight'] = (o, i, 1); s[p + name + '.bias'] = (o,)
    for name, (o, i) in {'fc1': (512, 1024), 'fc2': (256, 512), 'fc3': (k * k, 256)}.items():
        s[p + name + '.weight'] = (o, i); s[p + name + '.bias'] = (o,)
    for name, c in {'bn1': 64, 'bn2': 128, 'bn3': 1024, 'bn4': 512, 'bn5': 256}.items():
        s[p + name] = c
    return s


def model_shapes(kind, n_in, n_out):
    """Parameter / buffer names and shapes of PointNetCls ('cls') / PointNetSeg ('seg')."""
    s = {}
    s.update(_stn_shapes('feat.stn.', n_in, 3))
    for name, (o, i) in {'conv1': (64, n_in), 'conv2': (128, 64), 'conv3': (1024, 128)}.items():
        s['feat.' + name + '.weight'] = (o, i, 1); s['feat.' + name + '.bias'] = (o,)
    for name, c in {'bn1': 64, 'bn2': 128, 'bn3': 1024}.items():
        s['feat.' + name] = c
    s.update(_stn_shapes('feat.fstn.', 64, 64))
    if kind == 'cls':
        for name, (o, i) in {'fc1': (512, 1024), 'fc2': (256, 512), 'fc3': (n_out, 256)}.items():
            s[name + '.weight'] = (o, i); s[name + '.bias'] = (o,)
        s['bn1'] = 512; s['bn2'] = 256
    else:
        for name, (o, i) in {'conv1': (512, 1088), 'conv2': (256, 512), 'conv3': (128, 256), 'conv4': (n_out, 128)}.items():
            s[name + '.weight'] = (o, i, 1); s[name + '.bias'] = (o,)
        s['bn1'] = 512; s['bn2'] = 256; s['bn3'] = 128
    return s


def make_state_dict(kind, n_in, n_out, seed=0, prefix='', gain=1.6):
    """Seeded synthetic checkpoint with non-trivial BN statistics so folding is exercised.
    Weights ~ U(-gain/sqrt(fan_in), gain/sqrt(fan_in)); gain=1 is torch's default init, gain=1.6 keeps
    activations O(1) through the stack so logits are O(1-10) rather than ~0."""
    import torch
    rng = np.random.default_rng(seed)
    shapes = model_shapes(kind, n_in, n_out)
    sd = {}
    for name, shp in shapes.items():
        if isinstance(shp, int):
            c = shp
            sd[prefix + name + '.weight'] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
            sd[prefix + name + '.bias'] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
            sd[prefix + name + '.running_mean'] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
            sd[prefix + name + '.running_var'] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
            sd[prefix + name + '.num_batches_tracked'] = torch.tensor(100, dtype=torch.long)
        else:
            fan_in = shp[1] if len(shp) > 1 else shapes[name.replace('.bias', '.weight')][1]
            b = gain / np.sqrt(fan_in)
            sd[prefix + name] = torch.from_numpy(rng.uniform(-b, b, shp).astype(np.float32))
    return sd


def seeded_like(state_dict, seed=0, gain=1.6):
    """A seeded synthetic checkpoint for ANY module of the pointnet2 family, from the names / shapes of its own state_dict (key
    order = the module's): BatchNorm statistics as in make_state_dict, weights ~ U(+-gain/sqrt(fan_in)).  Used for the standalone
    building blocks (STN3d / STNkd / PointNetEncoder with the reference's other constructor arguments)."""
    import torch
    rng = np.random.default_rng(seed)
    out = {}
    fan = {}
    for name, t in state_dict.items():
        shp = tuple(t.shape)
        leaf = name.rsplit('.', 1)[-1]
        is_bn = (name.rsplit('.', 1)[0] + '.running_var') in state_dict
        if leaf == 'num_batches_tracked':
            out[name] = torch.tensor(100, dtype=torch.long)
        elif is_bn:
            v = rng.uniform(0.5, 1.5, shp) if leaf in ('weight', 'running_var') else rng.normal(0, 0.1, shp)
            out[name] = torch.from_numpy(v.astype(np.float32))
        else:
            if leaf == 'weight':
                fan[name.rsplit('.', 1)[0]] = int(np.prod(shp[1:]))
            b = gain / np.sqrt(fan[name.rsplit('.', 1)[0]])
            out[name] = torch.from_numpy(rng.uniform(-b, b, shp).astype(np.float32))
    return out


# ---------------------------------------------------------------------------------------------
# objects, scenes
# ---------------------------------------------------------------------------------------------
def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def nut_surface(n, rng, r_in=0.004, r_out=0.008, height=0.006):
    """Points + outward normals on a nut-like annular prism (object frame, metres)."""
    a_out = 2 * np.pi * r_out * height
    a_in = 2 * np.pi * r_in * height
    a_cap = np.pi * (r_out ** 2 - r_in ** 2)
    probs = np.array([a_out, a_in, a_cap, a_cap]); probs /= probs.sum()
    which = rng.choice(4, size=n, p=probs)
    th = rng.uniform(0, 2 * np.pi, n)
    z = rng.uniform(-height / 2, height / 2, n)
    r = np.sqrt(rng.uniform(r_in ** 2, r_out ** 2, n))
    pts = np.zeros((n, 3)); nrm = np.zeros((n, 3))
    c, s = np.cos(th), np.sin(th)
    m = which == 0
    pts[m] = np.stack([r_out * c[m], r_out * s[m], z[m]], 1); nrm[m] = np.stack([c[m], s[m], 0 * z[m]], 1)
    m = which == 1
    pts[m] = np.stack([r_in * c[m], r_in * s[m], z[m]], 1); nrm[m] = np.stack([-c[m], -s[m], 0 * z[m]], 1)
    m = which == 2
    pts[m] = np.stack([r[m] * c[m], r[m] * s[m], 0 * z[m] + height / 2], 1); nrm[m] = [0, 0, 1]
    m = which == 3
    pts[m] = np.stack([r[m] * c[m], r[m] * s[m], 0 * z[m] - height / 2], 1); nrm[m] = [0, 0, -1]
    return pts, nrm


def screw_surface(n, rng, r=0.0025, length=0.04, head_r=0.005, head_len=0.01):
    """'hnm'/'screw'-like: cylinder 5 x 40 mm with a 10 mm head."""
    a_shaft = 2 * np.pi * r * length
    a_head = 2 * np.pi * head_r * head_len
    a_top = np.pi * head_r ** 2
    probs = np.array([a_shaft, a_head, a_top]); probs /= probs.sum()
    which = rng.choice(3, size=n, p=probs)
    th = rng.uniform(0, 2 * np.pi, n); c, s = np.cos(th), np.sin(th)
    pts = np.zeros((n, 3)); nrm = np.zeros((n, 3))
    m = which == 0
    x = rng.uniform(0, length, n)
    pts[m] = np.stack([x[m], r * c[m], r * s[m]], 1); nrm[m] = np.stack([0 * x[m], c[m], s[m]], 1)
    m = which == 1
    x = rng.uniform(-head_len, 0, n)
    pts[m] = np.stack([x[m], head_r * c[m], head_r * s[m]], 1); nrm[m] = np.stack([0 * x[m], c[m], s[m]], 1)
    m = which == 2
    rr = head_r * np.sqrt(rng.uniform(0, 1, n))
    pts[m] = np.stack([0 * rr[m] - head_len, rr[m] * c[m], rr[m] * s[m]], 1); nrm[m] = [-1, 0, 0]
    return pts, nrm


# mixed-category bins: object k belongs to category MIXED_BINS[kind][k % 3] ('hnm' and 'screw' share the synthetic bolt geometry,
# SURVEY.md §8(d); they differ in symmetry count and in the weights of their predicters)
MIXED_BINS = {'mixed': ['nut', 'screw', 'screw'], 'bin': ['nut', 'hnm', 'screw']}


def make_scene(n_objects, pts_per_object, seed=0, kind='nut'):
    """Clutter pile in the camera frame: objects at random SE(3) poses inside a 10x10x4 cm box at
    z in [0.55,0.75] m.  Returns list of dict(xyz (M,3) f64, normal (M,3) f64, pose 4x4)."""
    rng = np.random.default_rng(seed)
    objs = []
    for k in range(n_objects):
        kk = MIXED_BINS[kind][k % 3] if kind in MIXED_BINS else kind
        p, n = (nut_surface if kk == 'nut' else screw_surface)(pts_per_object, rng)
        R = random_rotation(rng)
        t = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(0.55, 0.59) + 0.16 * rng.uniform()])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        objs.append({'xyz': p @ R.T + t, 'normal': n @ R.T, 'pose': T, 'kind': kk})
    return objs


def make_candidates(obj, n, rng, hand_depth=0.04, init_bite=0.005):
    """Grasp candidates for one object (camera frame 4x4 float64), mirroring the cone sampler
    (dexnet/grasping/grasp_sampler.py:165-168,269-289): approach axis (x) within 60 deg of -normal at a
    random surface point, random in-plane rotation, standoff U(0, hand_depth)."""
    xyz, nrm = obj['xyz'], obj['normal']
    idx = rng.integers(0, len(xyz), n)
    poses = np.zeros((n, 4, 4)); poses[:, 3, 3] = 1
    for i, k in enumerate(idx):
        a0 = -nrm[k] / np.linalg.norm(nrm[k])
        # random direction within 60 deg of a0
        while True:
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            if d @ a0 >= np.cos(np.pi / 3):
                break
        ref = np.array([0., 0., 1.]) if abs(d[2]) < 0.9 else np.array([1., 0., 0.])
        y = np.cross(ref, d); y /= np.linalg.norm(y)
        z = np.cross(d, y)
        ang = rng.uniform(0, np.pi)
        y2 = np.cos(ang) * y + np.sin(ang) * z
        z2 = np.cross(d, y2)
        dist = rng.uniform(0, hand_depth)
        poses[i, :3, 0] = d; poses[i, :3, 1] = y2; poses[i, :3, 2] = z2
        poses[i, :3, 3] = xyz[k] + init_bite * d + d * dist
    return poses


# ---------------------------------------------------------------------------------------------
# gripper
# ---------------------------------------------------------------------------------------------
def box_mesh(lo, hi):
    lo = np.asarray(lo, float); hi = np.asarray(hi, float)
    v = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
                  [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], dtype=np.int32)
    return v, f


def _merge(meshes):
    vs, fs, off = [], [], 0
    for v, f in meshes:
        vs.append(v); fs.append(f + off); off += len(v)
    return np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.int32)


def make_gripper(opening=0.04, finger_len=0.04, finger_w=0.01, finger_h=0.02, subdivisions=0):
    """Synthetic parallel-jaw gripper in its base frame (x = approach, y = closing direction):
    palm box + two finger boxes (open mesh, 36 triangles) and the same with the inter-finger volume
    filled (enclosed mesh, 48 triangles).  gripper_in_grasp puts the finger tips 5 mm past the grasp centre."""
    half = opening / 2
    palm = box_mesh([-0.03, -(half + finger_w) - 0.005, -0.015], [0.0, (half + finger_w) + 0.005, 0.015])
    f1 = box_mesh([0.0, half, -finger_h / 2], [finger_len, half + finger_w, finger_h / 2])
    f2 = box_mesh([0.0, -half - finger_w, -finger_h / 2], [finger_len, -half, finger_h / 2])
    fill = box_mesh([0.0, -half, -finger_h / 2], [finger_len, half, finger_h / 2])
    V, F = _merge([palm, f1, f2])
    Ve, Fe = _merge([palm, f1, f2, fill])
    if subdivisions:                       # same surfaces, 4^subdivisions times the triangles
        V, F = subdivide(V, F, subdivisions); Ve, Fe = subdivide(Ve, Fe, subdivisions)
    gripper_in_grasp = np.eye(4); gripper_in_grasp[0, 3] = -(finger_len - 0.005)
    return {'vertices': V, 'faces': F, 'enclosed_vertices': Ve, 'enclosed_faces': Fe,
            'gripper_in_grasp': gripper_in_grasp, 'hand_depth': finger_len, 'init_bite': 0.005,
            'diameter': float(np.linalg.norm(V.max(0) - V.min(0)))}


def subdivide(V, F, times):
    """Loop-free 1:4 midpoint subdivision of a triangle mesh, `times` times (36 box triangles x 4^4 = 9,216: a gripper mesh of the
    size real CAD exports have; the surface, and therefore every collision verdict up to rounding, is unchanged)."""
    V = np.asarray(V, dtype=np.float32); F = np.asarray(F, dtype=np.int32)
    for _ in range(times):
        a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
        n0 = len(V)
        mids = np.stack([(a + b) / 2, (b + c) / 2, (c + a) / 2], axis=1).reshape(-1, 3).astype(np.float32)
        i = n0 + 3 * np.arange(len(F))
        F = np.concatenate([np.stack([F[:, 0], i, i + 2], 1), np.stack([i, F[:, 1], i + 1], 1), np.stack([i + 2, i + 1, F[:, 2]], 1),
                            np.stack([i, i + 1, i + 2], 1)]).astype(np.int32)
        V = np.concatenate([V, mids])
    return V, F


def background_points(objs, k, gripper_diameter):
    """Scene points within gripper_diameter/2 of object k, minus the object itself
    (run_grasp_simulation.py:131-135, without the ray-cast occupancy fill)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(objs[k]['xyz'])
    others = [o['xyz'] for i, o in enumerate(objs) if i != k]
    if not others:
        return np.zeros((0, 3))
    pts = np.concatenate(others)
    d, _ = tree.query(pts)
    return pts[d <= gripper_diameter / 2]
