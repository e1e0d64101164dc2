"""Device form of the reference's PointConeGraspSampler candidate generation
(dexnet/grasping/grasp_sampler.py:155-298): per sampled surface point a local frame from the neighbours' normals, then the
fan-out over approach directions x in-plane rotations x approach depths; optionally centred between the fingers.
Returns poses only; filtering is `catgrasp_amd.my_cpp.filterGraspPose`."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import _p, _stream, check

_c_int = ctypes.c_int
_c_long = ctypes.c_long


def cone_grasp_poses(points_for_sample, normals_for_sample, sample_ids, sphere_pts, r_ball, hand_depth, init_bite, approach_step,
                     inplane_step_deg=30.0, center_ob_between_gripper=False, device=None, return_tensor=False, info=None):
    """-> (K*(1+S*n_rot)*n_depth, 4, 4) float64 grasp poses in the order of the reference's nested loops
    (sample point, rotation, depth).  `r_ball` is the starting ball radius (3 x cloud resolution in the reference, :160);
    the reference's persistent doubling of r_ball when a point has no usable neighbour (:243-247) is reproduced."""
    if device is None:
        if not torch.cuda.is_available():
            raise L.CatgraspAmdError('catgrasp_amd.grasp_sampler needs a HIP device (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())
    pts = torch.from_numpy(np.ascontiguousarray(np.asarray(points_for_sample, dtype=np.float64).reshape(-1, 3))).to(device)
    nrm = torch.from_numpy(np.ascontiguousarray(np.asarray(normals_for_sample, dtype=np.float64).reshape(-1, 3))).to(device)
    ids = torch.from_numpy(np.ascontiguousarray(np.asarray(sample_ids, dtype=np.int32).reshape(-1))).to(device)
    sph = torch.from_numpy(np.ascontiguousarray(np.asarray(sphere_pts, dtype=np.float64).reshape(-1, 3))).to(device)
    P, K, S = pts.shape[0], ids.shape[0], sph.shape[0]
    n_rot = len(np.arange(0, 180, inplane_step_deg))
    n_depth = len(np.arange(0, hand_depth, approach_step))
    total = K * (1 + S * n_rot) * n_depth
    out = torch.empty((total, 16), dtype=torch.float64, device=device)
    if total == 0:
        return out.view(0, 4, 4) if return_tensor else out.view(0, 4, 4).cpu().numpy()
    lib = L.lib()
    dbl = torch.zeros((K,), dtype=torch.int32, device=device)
    check(lib.cg_cone_frames(_p(pts), _p(nrm), _c_int(P), _p(ids), _c_int(K), None, ctypes.c_double(float(r_ball)), _c_int(0), _p(dbl), None,
                             _stream()), 'cg_cone_frames')
    # the reference mutates self.params['r_ball'] while it walks the points in order: radius of point k = r0 * 2^(max doublings so far)
    radii = float(r_ball) * np.power(2.0, np.maximum.accumulate(dbl.cpu().numpy().astype(np.float64)))
    d_r = torch.from_numpy(radii).to(device)
    if info is not None:
        info['radii'] = radii               # ball radius used at each sample point (after the reference's persistent doublings)
    frames = torch.empty((K, 9), dtype=torch.float64, device=device)
    check(lib.cg_cone_frames(_p(pts), _p(nrm), _c_int(P), _p(ids), _c_int(K), _p(d_r), ctypes.c_double(float(r_ball)), _c_int(1), None, _p(frames),
                             _stream()), 'cg_cone_frames')
    check(lib.cg_cone_poses(_p(pts), _p(ids), _p(frames), _c_int(K), _p(sph), _c_int(S), _c_int(n_rot), ctypes.c_double(float(inplane_step_deg)),
                            _c_int(n_depth), ctypes.c_double(float(approach_step)), ctypes.c_double(float(init_bite)), _p(out), _stream()),
          'cg_cone_poses')
    if center_ob_between_gripper:
        check(lib.cg_center_grasps(_p(out), _c_long(total), _p(pts), _c_int(P), _stream()), 'cg_center_grasps')
    out = out.view(total, 4, 4)
    return out if return_tensor else out.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# Class-level mirrors of the reference's samplers (dexnet/grasping/grasp_sampler.py:66-108, :155-222, :302-370): same
# constructor / sample_grasps signatures, same consumption of numpy's global RNG, candidate generation and filtering on the
# device.  They return lists of `ParallelJawPtGrasp3D` (an object with `.grasp_pose`, 4x4) like the reference.

class ParallelJawPtGrasp3D:
    """The fields of dexnet/grasping/grasp.py:113-166 this path reads: `.grasp_pose` (4x4, grasp in object / camera frame) and
    `.perturbation_score`.  Instances unpickled from the reference's grasp files keep every other attribute they were saved with."""

    def __init__(self, grasp_pose=None, perturbation_score=0.0):
        self.grasp_pose = None if grasp_pose is None else np.asarray(grasp_pose)
        self.perturbation_score = perturbation_score

    def get_grasp_pose_matrix(self):
        if self.grasp_pose is None:      # grasp.py:163-166 falls back to T_grasp_obj (autolab_core); the files of this pipeline carry the pose
            raise ValueError('grasp without grasp_pose (configuration-only grasps are not part of the scoring path)')
        return np.array(self.grasp_pose, copy=True)


class ReferenceObject:
    """Attribute bag for a pickled reference object whose class lives in a package that is not installed here (dexnet contacts,
    autolab_core transforms, meshpy / trimesh meshes): keeps the saved attributes, carries no behaviour."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:          # (dict state, slots state)
            state = {**(state[0] or {}), **(state[1] or {})}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['state'] = state


_REFERENCE_PACKAGES = ('dexnet', 'autolab_core', 'meshpy', 'trimesh', 'perception', 'visualization')
_reference_classes = {}
# The only non-reference globals a canonical-model / grasp-list pickle needs: numpy array reconstruction and plain containers.
# Anything else (os.system, builtins.eval, ...) is refused instead of resolved -- a crafted .pkl cannot run code through this loader.
_SAFE_GLOBALS = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
                 ('numpy._core.multiarray', 'scalar'), ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy.core.numeric', '_frombuffer'),
                 ('numpy._core.numeric', '_frombuffer'), ('collections', 'OrderedDict'), ('collections', 'defaultdict'),
                 ('builtins', 'list'), ('builtins', 'dict'), ('builtins', 'tuple'), ('builtins', 'set'), ('builtins', 'frozenset'),
                 ('builtins', 'int'), ('builtins', 'float'), ('builtins', 'bool'), ('builtins', 'str'), ('builtins', 'bytes'),
                 ('builtins', 'complex'), ('builtins', 'slice'), ('builtins', 'range'), ('builtins', 'bytearray'), ('copyreg', '_reconstructor'),
                 ('builtins', 'object')}


def _reference_unpickler(f):
    import pickle

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == 'dexnet.grasping.grasp' and name == 'ParallelJawPtGrasp3D':
                return ParallelJawPtGrasp3D
            if module.split('.')[0] in _REFERENCE_PACKAGES:
                key = (module, name)
                if key not in _reference_classes:
                    _reference_classes[key] = type(name, (ReferenceObject,), {'__module__': module})
                return _reference_classes[key]
            if (module, name) in _SAFE_GLOBALS:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f'load_reference_pickle: global {module}.{name} is not on the allowlist of this loader')
    return Unpickler(f)


def load_reference_pickle(path):
    """Read a gzip pickle written by the reference -- `data/object_models/{class}_canonical.pkl` (make_canonical.py:155-164: dict with
    'canonical_cloud', 'canonical_normals', 'canonical_affordance', 'canonical_grasps', 'transforms_to_nocs', 'obj_files'; read at
    run_grasp_simulation.py:706-707) or a `*_complete_grasp.pkl` grasp list (generate_grasp.py) -- without dexnet / autolab_core:
    grasps come back as this module's ParallelJawPtGrasp3D with all saved attributes, other reference objects as attribute bags.
    Stricter than the reference's plain pickle.load: only numpy reconstruction and plain containers are resolved besides the
    reference's own classes (which become inert attribute bags); any other global raises pickle.UnpicklingError."""
    import gzip
    with open(path, 'rb') as raw:
        gz = raw.read(2) == b'\x1f\x8b'
    with (gzip.open(path, 'rb') if gz else open(path, 'rb')) as f:
        return _reference_unpickler(f).load()


def load_canonical(path):
    """The canonical model of a category as NocsTransferGraspSampler and the affordance stage take it: the dict of
    `{class}_canonical.pkl` with 'canonical_grasps' as a python list."""
    canonical = dict(load_reference_pickle(path))
    for k in ('canonical_cloud', 'canonical_normals', 'canonical_affordance', 'canonical_grasps'):
        if k not in canonical:
            raise KeyError(f'{path}: not a canonical model file (no {k!r})')
    canonical['canonical_grasps'] = list(canonical['canonical_grasps'])
    return canonical


def hinter_sampling(min_n_pts, radius=1):
    """View-sphere sampling by icosahedron refinement (Hinterstoisser et al., BMVC 2008), enumerating vertices and faces in
    the order of Utils.py:293-360 so that an index subset drawn from the result selects the same directions.
    -> (pts (n,3), refinement level of each point)."""
    a, b, c = 0.0, 1.0, (1.0 + np.sqrt(5.0)) / 2.0
    pts = [(-b, c, a), (b, c, a), (-b, -c, a), (b, -c, a), (a, -b, c), (a, b, c), (a, -b, -c), (a, b, -c), (c, a, -b), (c, a, b),
           (-c, a, -b), (-c, a, b)]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    level = [0] * len(pts)
    ref_level = 0
    while len(pts) < min_n_pts:
        ref_level += 1
        midpoint = {}
        new_faces = []
        for face in faces:
            ids = list(face)
            for i in range(3):
                e = (min(face[i], face[(i + 1) % 3]), max(face[i], face[(i + 1) % 3]))
                if e not in midpoint:
                    midpoint[e] = len(pts)
                    pts.append((0.5 * (np.array(pts[e[0]]) + np.array(pts[e[1]]))).tolist())
                    level.append(ref_level)
                ids.append(midpoint[e])
            new_faces += [(ids[0], ids[3], ids[5]), (ids[3], ids[1], ids[4]), (ids[3], ids[4], ids[5]), (ids[5], ids[4], ids[2])]
        faces = new_faces
    pts = np.array(pts, dtype=np.float64)
    pts *= (radius / np.linalg.norm(pts, axis=1)).reshape(-1, 1)
    # output order (Utils.py:353-386): breadth-first over the mesh edges from the top vertex, every ring sorted by azimuth
    # (stable sort; a ring is gathered through a python set, whose iteration order for ints is deterministic)
    neighbours = {}
    for face in faces:
        for i in range(3):
            neighbours.setdefault(face[i], set()).update((face[(i + 1) % 3], face[(i + 2) % 3]))
    two_pi = 2.0 * np.pi
    order, done = [], [False] * len(pts)
    ring = [int(np.argmax(pts[:, 2]))]
    while len(order) != len(pts):
        ring = sorted(ring, key=lambda i: (np.arctan2(pts[i][1], pts[i][0]) + two_pi) % two_pi)
        nxt = []
        for i in ring:
            order.append(i)
            done[i] = True
            nxt += list(neighbours[i])
        ring = [i for i in set(nxt) if not done[i]]
    order = np.array(order)
    return pts[order], [level[i] for i in order]


def compute_cloud_resolution(pts, n_sample=100, device=None):
    """Utils.py:492-501: mean of the 10 smallest nearest-neighbour distances from `n_sample` randomly drawn points
    (np.random.choice, global RNG) to the rest of the cloud; the nearest-neighbour search runs on the device."""
    from .affordance import nearest_neighbor
    pts = np.asarray(pts, dtype=np.float64)
    ids = np.random.choice(len(pts), size=n_sample).astype(int)
    rest = np.ones(len(pts), dtype=bool)
    rest[ids] = False
    background = pts[rest]
    nn = nearest_neighbor(pts[ids], background, device).cpu().numpy()
    dists = np.linalg.norm(pts[ids] - background[nn], axis=1)
    return np.sort(dists[np.isfinite(dists)])[:10].mean()


class GraspSampler:
    """Base class (grasp_sampler.py:66-108): keeps the gripper and whatever sampler configuration keys are present."""

    def __init__(self, gripper, config=None):
        self.gripper = gripper
        self.config = dict(config) if config is not None else {}
        for key, attr in (('sampling_friction_coef', 'friction_coef'), ('num_cone_faces', 'num_cone_faces'),
                          ('grasp_samples_per_surface_point', 'num_samples'), ('min_contact_dist', 'min_contact_dist')):
            if key in self.config:
                setattr(self, attr, self.config[key])

    def _filter(self, grasp_poses, symmetry_tfs, nocs_pose, cam_in_world, ee_in_grasp, filter_approach_dir_face_camera, filter_ik,
                adjust_collision_pose, upper, lower, open_gripper_collision_pts, background_pts, verbose):
        from . import my_cpp
        I4 = np.eye(4)
        g = self.gripper
        gripper_in_grasp = np.linalg.inv(g.get_grasp_pose_in_gripper_base())
        return my_cpp.filterGraspPose(grasp_poses, list(symmetry_tfs), nocs_pose, I4, I4 if cam_in_world is None else cam_in_world,
                                      I4 if ee_in_grasp is None else ee_in_grasp, gripper_in_grasp, filter_approach_dir_face_camera, filter_ik,
                                      adjust_collision_pose, upper if upper is not None else [0] * 7, lower if lower is not None else [0] * 7,
                                      g.trimesh.vertices, g.trimesh.faces, g.trimesh_enclosed.vertices, g.trimesh_enclosed.faces,
                                      open_gripper_collision_pts, background_pts, 0.0005, verbose)


class PointConeGraspSampler(GraspSampler):
    def candidate_poses(self, points_for_sample, normals_for_sample, max_num_samples=200, n_sphere_dir=100, approach_step=0.003,
                        center_ob_between_gripper=False):
        """The candidate generation of sample_grasps (grasp_sampler.py:156-198) -> (n,4,4) float64 poses, before filtering."""
        points_for_sample = np.asarray(points_for_sample, dtype=np.float64)
        resolution = compute_cloud_resolution(points_for_sample)
        self.params = {'debug_vis': False, 'r_ball': resolution * 3}
        self.approach_step = approach_step
        sphere_pts = hinter_sampling(min_n_pts=1000, radius=1)[0]
        sphere_pts = sphere_pts / np.linalg.norm(sphere_pts, axis=-1).reshape(-1, 1)
        sphere_pts = sphere_pts[sphere_pts[:, 2] >= np.cos(60 * np.pi / 180)]
        rot_y_90 = np.array([[0.0, 0, 1], [0, 1, 0], [-1, 0, 0]])      # euler_matrix(0, pi/2, 0, 'sxyz'): cone axis z -> x (approach)
        sphere_pts = (rot_y_90 @ sphere_pts.T).T
        if sphere_pts.shape[0] > n_sphere_dir:
            sphere_pts = sphere_pts[np.random.choice(np.arange(len(sphere_pts)), size=n_sphere_dir, replace=False)]
        sample_ids = np.arange(len(points_for_sample))
        np.random.shuffle(sample_ids)
        sample_ids = sample_ids[:max_num_samples]
        # the reference re-seeds numpy's GLOBAL generator with its own first state word at every surface point
        # (`seed = np.random.get_state()[1][0]` :183, `np.random.seed(seed)` in sample_one_surface_point): nothing random is drawn
        # there, but the generator state a later predict_batch / RANSAC sees is the re-seeded one -- reproduce it
        seed = np.random.get_state()[1][0]
        if len(sample_ids):
            np.random.seed(seed)
        self.info = {}
        poses = cone_grasp_poses(points_for_sample, normals_for_sample, sample_ids, sphere_pts, self.params['r_ball'], self.gripper.hand_depth,
                                 self.gripper.init_bite, approach_step, center_ob_between_gripper=center_ob_between_gripper, info=self.info)
        if len(self.info.get('radii', [])):
            self.params['r_ball'] = float(self.info['radii'][-1])        # the reference keeps the grown radius (:243-247)
        return poses

    def sample_grasps(self, background_pts, points_for_sample, normals_for_sample, max_num_samples=200, n_sphere_dir=100, approach_step=0.003,
                      ee_in_grasp=None, cam_in_world=None, upper=None, lower=None, open_gripper_collision_pts=None,
                      center_ob_between_gripper=False, filter_ik=True, adjust_collision_pose=True, **kwargs):
        """grasp_sampler.py:156-222: cone candidates, then filterGraspPose with symmetry [I], nocs_pose I, approach-direction filter on."""
        poses = self.candidate_poses(points_for_sample, normals_for_sample, max_num_samples, n_sphere_dir, approach_step, center_ob_between_gripper)
        kept = self._filter(poses, [np.eye(4)], np.eye(4), cam_in_world, ee_in_grasp, True, filter_ik, adjust_collision_pose, upper, lower,
                            open_gripper_collision_pts, background_pts, True)
        return [ParallelJawPtGrasp3D(grasp_pose=p) for p in kept]


class NocsTransferGraspSampler(GraspSampler):
    def __init__(self, gripper, config, canonical, class_name, score_larger_than=0, max_n_grasp=None, center_ob_between_gripper=False):
        """grasp_sampler.py:302-327: keep canonical grasps above a score, the best `max_n_grasp`, optionally centred in y."""
        super().__init__(gripper, config)
        grasps = list(canonical['canonical_grasps'])
        self.class_name = class_name
        if score_larger_than > 0:
            grasps = [g for g in grasps if g.perturbation_score >= score_larger_than]
        if max_n_grasp is not None:
            grasps = sorted(grasps, key=lambda x: -x.perturbation_score)[:max_n_grasp]
        if center_ob_between_gripper:
            centred = []
            for g in grasps:
                ob_in_grasp = np.linalg.inv(g.get_grasp_pose_matrix())
                ob_in_grasp[1, 3] = 0
                centred.append(ParallelJawPtGrasp3D(np.linalg.inv(ob_in_grasp), g.perturbation_score))
            grasps = centred
        self.canonical = dict(canonical)
        self.canonical['canonical_grasps'] = grasps

    def sample_grasps(self, background_pts, open_gripper_collision_pts, normals_for_sample, nocs_pts, nocs_pose, cam_in_world, ee_in_grasp, upper,
                      lower, filter_approach_dir_face_camera=False, ik_func=None, symmetry_tfs=(np.eye(4),), filter_ik=True, **kwargs):
        """grasp_sampler.py:330-356: canonical grasps x symmetry transforms through nocs_pose, pose nudging on."""
        poses = [g.get_grasp_pose_matrix() for g in self.canonical['canonical_grasps']]
        if not poses:
            return []
        kept = self._filter(poses, symmetry_tfs, nocs_pose, cam_in_world, ee_in_grasp, filter_approach_dir_face_camera, filter_ik, True, upper,
                            lower, open_gripper_collision_pts, background_pts, False)
        return [ParallelJawPtGrasp3D(grasp_pose=p) for p in kept]


class CombinedGraspSampler(GraspSampler):
    def __init__(self, gripper, config, samplers=()):
        super().__init__(gripper, config)
        self.samplers = list(samplers)

    def sample_grasps(self, **kwargs):
        grasps_all = []
        for sampler in self.samplers:
            grasps_all += sampler.sample_grasps(**kwargs)
        return grasps_all
