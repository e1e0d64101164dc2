"""Drop-in for the reference's `predicter.GraspPredicter` / `predicter.NunocsPredicter`
(predicter.py:39-203) on MI355X: same constructor argument, attributes and method results, with the
per-candidate python transform loop (predicter.py:71-74), the chunk-of-200 forward (predicter.py:76-91)
and the NUNOCS decode (predicter.py:144-150) replaced by device kernels.

Extra keyword arguments (all optional) let a caller supply what the reference reads from
`artifacts/artifacts-<id>/` (config, normalizer, checkpoint) directly -- the artifacts are external
downloads that are not part of the reference repository.
"""
import os
import pickle

import numpy as np
import torch
import yaml

from . import engine, folding, ops, transforms

DEFAULT_GRASP_CFG = {'n_pts': 2048, 'input_channel': 6,
                     'classes': [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.01]}   # config_grasp.yml:9,14,15
DEFAULT_NUNOCS_CFG = {'n_pts': 8192, 'input_channel': 6, 'ce_loss_bins': 100}             # config_nunocs.yml:10,15,16


def load_state_dict(ckpt_dir):
    """Utils.load_model (Utils.py:135-148): accepts {'state_dict': ...} or a bare state dict and strips
    the DataParallel 'module.' prefix."""
    sd = torch.load(ckpt_dir, map_location='cpu', weights_only=False)
    if 'state_dict' in sd:
        sd = sd['state_dict']
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    assert len(sd) > 0
    return sd


def artifact_root():
    """The reference reads `<dir of predicter.py>/artifacts/artifacts-<id>/` (predicter.py:47-48).  Here the root is
    $CATGRASP_ARTIFACTS if set, else the directory that contains this package (the reference checkout when
    `catgrasp_amd/` is dropped into it)."""
    return os.environ.get('CATGRASP_ARTIFACTS', os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), 'artifacts'))


def _load_artifacts(artifact_dir, cfg_name, cfg, state_dict, normalizer):
    if cfg is None:
        with open(f'{artifact_dir}/{cfg_name}', 'r') as ff:
            cfg = yaml.safe_load(ff)
    cfg = dict(cfg)
    if normalizer is None and artifact_dir is not None and os.path.exists(f'{artifact_dir}/normalizer.pkl'):
        with open(f'{artifact_dir}/normalizer.pkl', 'rb') as ff:
            normalizer = pickle.load(ff)
    if normalizer is not None:
        cfg['mean'] = np.asarray(normalizer['mean'])
        cfg['std'] = np.asarray(normalizer['std'])
    if state_dict is None:
        state_dict = load_state_dict(f'{artifact_dir}/best_val.pth.tar')
    return cfg, state_dict


def _pin(shape, dtype):
    """Page-locked staging buffer of predict_batch (both directions).  A copy from PAGEABLE host memory is not asynchronous on
    ROCm whatever `non_blocking` says: the runtime stages it in order on the stream and blocks the calling thread until it has run,
    i.e. until the device has finished the chunk before -- everything the host does after it (the rest of the chunk's launches,
    the rows of the previous chunk) is then done with the device idle (measured: 20-30 ms of a 50,000-pose call, rocprofv3 kernel
    trace).  torch's host allocator caches these blocks, so repeated calls of the same shape pay the page-locking once."""
    return torch.empty(shape, dtype=dtype, pin_memory=True)


_PIN_LIMIT = 512 << 20       # per staging buffer; a larger chunk of swap partners (a > 30k-point cloud at 16,384 poses) goes pageable


# Chunks of up to this many poses draw their resampling rows entirely on the host (numpy stream replay + swap chain, ~6 us per row)
# instead of sending swap partners to the device: a swap chain is a dependent sequence wherever it runs, one lane of the device
# needs ~250 us for it however few rows there are, so the device only wins when it runs thousands of chains side by side.
_HOST_ROWS_MAX = 32

_WORKER = []


def _draw_worker():
    """The one thread that replays numpy's global stream ahead of the device, shared by every predict_batch call of the process (the
    stream is global anyway): starting a thread per call is 0.1 ms of a 1.3 ms small call."""
    if not _WORKER:
        from concurrent.futures import ThreadPoolExecutor
        _WORKER.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix='catgrasp-numpy-stream'))
    return _WORKER[0]


_AHEAD = []


def draw_ahead_worker():
    """A second thread for draws made AHEAD of numpy's global stream (pipeline.evaluate_objects: the next object's NUNOCS-stage draws
    while the stream worker above is still handing the current object's resampling rows to the device)."""
    if not _AHEAD:
        from concurrent.futures import ThreadPoolExecutor
        _AHEAD.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix='catgrasp-numpy-ahead'))
    return _AHEAD[0]


def _event():
    ev = torch.cuda.Event(); ev.record()
    return ev


def _device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError('catgrasp_amd predicters need a HIP device (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class _TransformOnly:
    """Stand-in for the `.dataset` attribute of the reference predicters (predicter.py:60,127): keeps cfg /
    phase; the transform itself runs on the device inside predict*()."""

    def __init__(self, cfg, phase, class_name=None):
        self.cfg = cfg
        self.phase = phase
        self.class_name = class_name


class GraspPredicter:
    class_name_to_artifact_id = {'nut': 47, 'hnm': 51, 'screw': 50}          # predicter.py:41-45

    def __init__(self, class_name, artifact_dir=None, cfg=None, state_dict=None, normalizer=None, device=None,
                 chunk=16384):
        self.class_name = class_name
        if artifact_dir is None and (cfg is None or state_dict is None):
            artifact_dir = f"{artifact_root()}/artifacts-{self.class_name_to_artifact_id[class_name]}"
            print('GraspPredicter artifact_dir', artifact_dir)
        self.cfg, sd = _load_artifacts(artifact_dir, 'config_grasp.yml', cfg, state_dict, normalizer)
        self.device = _device(device)
        self.dataset = _TransformOnly(self.cfg, 'test', class_name)
        from .pointnet2 import PointNetCls
        self.model = PointNetCls(n_in=self.cfg['input_channel'], n_out=len(self.cfg['classes']) - 1)
        self.model.load_state_dict(sd)
        self.model.to(self.device).eval()
        self._W = folding.prepare_cls(sd, self.device)
        self._mean, self._inv_std = transforms.normalizer_device(self.cfg, self.device)
        self.chunk = int(chunk)
        self.rng = os.environ.get('CATGRASP_AMD_RNG', 'numpy')

    # ---- device-resident API (what bench.py and the multi-GPU path use) ----
    def upload_cloud(self, data):
        return transforms.DeviceCloud(data['cloud_xyz'], data['cloud_normal'], self.device)

    def score_on_device(self, cloud_xyz, cloud_normal, ids, pose_inv):
        """cloud_xyz/normal (M,3) f32 cuda; ids (G,n_pts) i32 cuda, or a callable ids(s, e) -> (e-s,n_pts) i32 cuda that is asked for
        each chunk of candidates in order, exactly once; pose_inv (G,12) f32 cuda.
        -> probs (G,C), label (G) i32, confidence (G), p_G (G) cuda tensors."""
        G = pose_inv.shape[0]
        guard = engine.current_precision() in engine.HALF_MODES
        id_chunks = {}      # chunks of a callable id source, kept ONLY under the half-range guard (a tripped chunk is scored again)

        def ids_of(s, e):
            if not callable(ids):
                return ids[s:e]
            if s in id_chunks:
                return id_chunks[s]
            c = ids(s, e)
            if guard:
                id_chunks[s] = c
            return c
        C = len(self.cfg['classes']) - 1
        logits = torch.empty((G, C), dtype=torch.float32, device=self.device)
        # chunk boundaries: uniform, or -- for an id source that is produced on the host while the device works (ids.ramp) -- a
        # short first chunk doubling up to the full size, so that the device starts after a fraction of the first draw
        sizes = [min(self.chunk, r) for r in getattr(ids, 'ramp', ())]
        bounds, s = [], 0
        while s < G:
            e = min(G, s + (sizes.pop(0) if sizes else self.chunk))
            bounds.append((s, e)); s = e
        if hasattr(ids, 'plan'):
            ids.plan(bounds)
        starts = [b[0] for b in bounds]
        ends = dict(bounds)
        status = engine.new_status(self.device, len(starts)) if guard else None      # one range word per chunk

        def run(s, st):
            e = ends[s]
            x = ops.build_grasp_input(cloud_xyz, cloud_normal, ids_of(s, e), pose_inv[s:e], self._mean, self._inv_std)
            logits[s:e] = engine.cls_forward(self._W, x, st)[0]
        for k, s in enumerate(starts):
            run(s, status[k:k + 1] if guard else None)
        if guard and G:
            bits = status.cpu().numpy()          # ONE read-back per call; chunks that left the half range are re-run with bf16 pieces
            if bits.any():
                engine.warn_range(int(np.bitwise_or.reduce(bits)))
                with engine.precision('bf16x3'):
                    for k in np.nonzero(bits)[0]:
                        run(starts[k], None)
        return ops.softmax_pg(logits)

    def _numpy_id_chunks(self, n_valid, n_pts, G):
        """ids(s, e) for score_on_device: numpy's global stream replayed in C (transforms.NumpyChoiceStream), one chunk ahead on a
        worker thread -- the draw of chunk k+1 overlaps the device scoring chunk k (the C call releases the GIL).  For the usual
        replace=False draw the host only extracts the swap partners from the stream (the part that is sequential) and the
        permutation's swap chain runs on the device (ops.apply_shuffle_rows): same rows, same generator state afterwards."""
        stream = transforms.NumpyChoiceStream(n_valid, n_pts)
        pool = _draw_worker()
        chunk, dev = self.chunk, self.device
        pending, plan, order = {}, {}, {}
        on_device = stream.on_device_chain
        draw = stream.draw_partners if on_device else stream.draw
        width, dtype = (stream.partner_stride, torch.uint16) if on_device else (n_pts, torch.int32)
        ring, uploaded = [None, None], [None, None]     # two page-locked staging buffers, chunk k goes through ring[k & 1]

        def set_plan(bounds):
            assert not pending, 're-planning with draws in flight'
            plan.clear(); plan.update(dict(bounds))
            order.clear(); order.update({s: k for k, (s, _) in enumerate(bounds)})
            ring[:] = [None, None]; uploaded[:] = [None, None]      # staging buffers are sized for ONE plan
            rows = max(e - s for s, e in bounds)
            if rows * width * (2 if on_device else 4) <= _PIN_LIMIT:
                ring[:] = [_pin((rows, width), dtype) for _ in range(min(2, len(bounds)))] + [None] * (2 - min(2, len(bounds)))

        def task(k, count):
            if on_device and count <= _HOST_ROWS_MAX:      # a few poses: whole rows on the host (see _HOST_ROWS_MAX)
                return torch.from_numpy(stream.draw(count))
            buf = ring[k & 1]
            if buf is None:
                return torch.from_numpy(draw(count))
            if uploaded[k & 1] is not None:
                uploaded[k & 1].synchronize()           # chunk k-2 has left this buffer
            draw(count, out=buf[:count].numpy())
            return buf[:count]

        def submit(s):
            if s in plan and s not in pending:
                pending[s] = pool.submit(task, order[s], plan[s] - s)

        def ids(s, e):
            if not plan:
                set_plan([(a, min(G, a + chunk)) for a in range(0, G, chunk)])
            assert plan.get(s) == e, 'chunks must be asked for in the planned order'
            if s not in pending and e - s <= _HOST_ROWS_MAX:      # a few rows: drawn right here, a thread hand-over costs more than the draw
                host = task(order[s], e - s)
            else:
                submit(s)
                host = pending.pop(s).result()
            submit(e)                                   # the next chunk is drawn while this one is uploaded and scored
            up = host.to(dev, non_blocking=True)
            uploaded[order[s] & 1] = _event()
            return ops.apply_shuffle_rows(up, n_valid, n_pts) if host.dtype == torch.uint16 else up

        ids.ramp = (2048, 4096, 8192)
        ids.plan = set_plan

        def close():
            for f in pending.values():
                f.result()
            stream.close()
        ids.close = close
        return ids

    # ---- reference API ----
    def _chunk_plan(self, G, id_source):
        """Chunk boundaries of a predict_batch call: full chunks of self.chunk candidates behind a ramp that starts at 1,024 and grows by
        1.5x, so that the device starts after a fraction of the host-side work of the first chunk (pose conversion, numpy-stream draw)
        and the host stays AHEAD during the ramp: the exact-f32 network takes ~12 us per candidate, the replay of numpy's stream ~6 us,
        so a doubling ramp is a dead heat (chunk k+1 is drawn in exactly the time chunk k is scored) and any slower host core stalls
        the device at every step of it; 1.5x leaves 25 % slack."""
        sizes = [min(self.chunk, r) for r in (1024, 1536, 2304, 3456, 5184, 7776, 11664)]
        bounds, s = [], 0
        while s < G:
            e = min(G, s + (sizes.pop(0) if sizes else self.chunk))
            bounds.append((s, e)); s = e
        if hasattr(id_source, 'plan'):
            id_source.plan(bounds)
        return bounds

    @staticmethod
    def _poses_f64(grasp_poses, s, e, out=None):
        """grasp_poses[s:e] (list of 4x4 arrays / nested lists, or an (G,4,4) array) -> contiguous float64 (e-s, 16), written into
        `out` (a staging buffer of that shape) when given."""
        part = grasp_poses[s:e]
        if isinstance(part, np.ndarray):
            P = np.ascontiguousarray(part, dtype=np.float64)
        elif len(part) and isinstance(part[0], np.ndarray) and part[0].dtype == np.float64 and part[0].shape == (4, 4):
            P = np.concatenate(part)                     # 1.4x faster than np.asarray on a list of small arrays
        else:
            P = np.asarray(part, dtype=np.float64)
        if P.size != (e - s) * 16:
            raise ValueError(f'grasp_poses[{s}:{e}] are not 4x4 matrices')
        P = P.reshape(-1, 16)
        if out is None:
            return P
        out[...] = P
        return out

    def predict_batch(self, data, grasp_poses, ids=None, rng=None):
        """predicter.py:67-94.  Returns [[pred_label, confidence, probs(10,) float32], ...] per grasp pose.
        `ids` (G,n_pts): explicit resample indices into the z>=0.1 filtered cloud.  Without them the per-pose resampling
        draw of GraspDataset.transform (dataset_grasp.py:72-73) comes from
          rng='numpy'  (default; $CATGRASP_AMD_RNG): numpy's GLOBAL generator, consumed exactly like the reference's one
                       np.random.choice per pose -- seeding numpy reproduces the reference's draws.  The stream is replayed in C one
                       chunk ahead of the device (the rejection sampling of the stream is inherently sequential: ~6 us per pose on one
                       host core; the permutation swap chains run on the device);
          rng='device': the same distribution drawn by a counter-based generator on the device (cg_draw_resample_ids; seeded
                       from one draw of numpy's global generator, so it is still reproducible under np.random.seed) -- no
                       host loop and no 8 KB/pose upload.
        The call is a pipeline over chunks of candidates: while the device scores chunk k, the calling thread converts the poses of
        chunk k+1 (the float64 matrices are uploaded as they are and inverted on the device, cg_pose_inverse_rows_f64) and turns the
        results of chunk k-1 into the reference's python rows; only the last chunk's rows are built with the device idle."""
        with torch.no_grad():
            G = len(grasp_poses)
            if G == 0:
                return []
            cloud = self.upload_cloud(data)
            n_pts = self.cfg['n_pts']
            rng_state = None
            if ids is None:
                rng = rng or self.rng
                if rng == 'device':
                    ids_d = transforms.draw_ids_device(cloud.n, n_pts, G, self.device, seed=int(np.random.randint(0, 2 ** 31)))
                elif rng == 'numpy':
                    self._poses_f64(grasp_poses, 0, min(G, 64))      # the usual malformed pose list fails before the generator is touched
                    rng_state = np.random.get_state()                # ... and a failure further down puts the generator back (below)
                    ids_d = self._numpy_id_chunks(cloud.n, n_pts, G)
                else:
                    raise ValueError(f"rng must be 'numpy' or 'device', not {rng!r}")
            else:
                ids = np.ascontiguousarray(ids, dtype=np.int32)
                if ids.shape != (G, n_pts):
                    raise ValueError(f'ids shape {ids.shape} != {(G, n_pts)}')
                if ids.size and (ids.min() < 0 or ids.max() >= cloud.n):      # numpy indexing in the reference raises too
                    raise IndexError(f'resample index out of range for a cloud of {cloud.n} valid points')
                ids_d = torch.from_numpy(ids).to(self.device)
            try:
                with _gc_paused():
                    return self._predict_chunks(cloud, grasp_poses, ids_d, G)
            except BaseException:
                try:
                    if hasattr(ids_d, 'close'):
                        ids_d.close(); ids_d.close = lambda: None
                except BaseException:            # a draw still in flight on the worker failed too: the caller gets the FIRST error
                    ids_d.close = lambda: None
                finally:
                    if rng_state is not None:    # the worker had drawn ahead of the failing chunk: a failed call consumes nothing
                        np.random.set_state(rng_state)     # (the reference's loop would have consumed its draws: INTEGRATION.md)
                raise
            finally:
                if hasattr(ids_d, 'close'):
                    ids_d.close()

    def _predict_chunks(self, cloud, grasp_poses, ids_d, G):
        bounds = self._chunk_plan(G, ids_d)
        C = len(self.cfg['classes']) - 1
        guard = engine.current_precision() in engine.HALF_MODES
        bad = torch.zeros((1,), dtype=torch.int32, device=self.device)
        stage = {'probs': _pin((G, C), torch.float32), 'label': _pin((G,), torch.int32), 'conf': _pin((G,), torch.float32),
                 'flags': _pin((len(bounds), 2), torch.int32).zero_(), 'poses': _pin((G, 16), torch.float64)}
        probs_h, label_h, conf_h, flags_h, poses_h = (stage[k].numpy() for k in ('probs', 'label', 'conf', 'flags', 'poses'))
        rows, pending = [], []

        def score(s, e, idc, status, bad_flag):
            """queue one chunk: pose upload + inverse, input transform, network, softmax, asynchronous copies into the staging buffers"""
            self._poses_f64(grasp_poses, s, e, out=poses_h[s:e])
            pinv = ops.pose_inverse_rows_f64(stage['poses'][s:e].to(self.device, non_blocking=True), cloud.center, bad_flag)
            x = ops.build_grasp_input(cloud.xyz, cloud.normal, idc, pinv, self._mean, self._inv_std)
            probs, label, conf, _ = ops.softmax_pg(engine.cls_forward(self._W, x, status)[0])
            stage['probs'][s:e].copy_(probs, non_blocking=True); stage['label'][s:e].copy_(label, non_blocking=True)
            stage['conf'][s:e].copy_(conf, non_blocking=True)

        def launch(k, s, e):
            idc = ids_d(s, e) if callable(ids_d) else ids_d[s:e]
            st = engine.new_status(self.device) if guard else None
            score(s, e, idc, st, bad)
            stage['flags'][k, 0:1].copy_(bad, non_blocking=True)
            if st is not None:
                stage['flags'][k, 1:2].copy_(st, non_blocking=True)
            return _event(), (idc if guard else None)        # the id chunk is kept only while a range re-run is still possible

        def drain(k, s, e, ev, idc):
            ev.synchronize()
            if flags_h[k, 0]:
                ops.raise_bad_poses(flags_h[k, 0])
            if guard and flags_h[k, 1]:                 # this chunk left the half range: score it again with bf16 pieces (rare)
                engine.warn_range(int(flags_h[k, 1]))
                with engine.precision('bf16x3'):
                    score(s, e, idc, None, None)
                _event().synchronize()
            pr = np.array(probs_h[s:e])                  # own copy: the rows below are views of it, the staging buffer is reused
            if not np.isfinite(pr).all():
                raise FloatingPointError('grasp-Q probabilities are not finite (non-finite weights or activations beyond float32)')
            # [label, confidence, probs row] per pose like predicter.py:87-91
            rows.extend(map(list, zip(label_h[s:e].copy(), conf_h[s:e].copy(), pr)))

        for k, (s, e) in enumerate(bounds):
            ev, idc = launch(k, s, e)
            pending.append((k, s, e, ev, idc))
            if len(pending) > 1:                         # chunk k is queued: turn chunk k-1 into rows while the device works on k
                drain(*pending.pop(0))
        while pending:
            drain(*pending.pop(0))
        return rows


class _gc_paused:
    """The rows of a predict_batch call are ~4 container objects per pose; every few hundred of them CPython's generational collector
    walks the young generations and, now and then, every tracked object of the process -- measured on the GPU box as 30-80 ms stalls
    of the chunk pipeline in one call out of three (profiles/r3_predict_batch_api.json).  None of these objects can be part of a
    cycle, so collection is held off for the duration of the call and the caller's setting restored."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self._was:
            import gc
            gc.enable()


class NunocsPredicter:
    class_name_to_artifact_id = {'nut': 78, 'hnm': 73, 'screw': 76}          # predicter.py:101-105

    def __init__(self, class_name, artifact_dir=None, cfg=None, state_dict=None, normalizer=None, device=None, align_fn=None,
                 ransac_sampling='reference'):
        self.class_name = class_name
        if self.class_name == 'nut':                                         # predicter.py:106-114
            self.min_scale = [0.005, 0.005, 0.001]
            self.max_scale = [0.05, 0.05, 0.05]
        else:
            self.min_scale = [0.005, 0.005, 0.005]
            self.max_scale = [0.15, 0.05, 0.05]
        if artifact_dir is None and (cfg is None or state_dict is None):
            artifact_dir = f"{artifact_root()}/artifacts-{self.class_name_to_artifact_id[class_name]}"
            print('NunocsPredicter artifact_dir', artifact_dir)
        self.cfg, sd = _load_artifacts(artifact_dir, 'config_nunocs.yml', cfg, state_dict, normalizer)
        self.device = _device(device)
        self.dataset = _TransformOnly(self.cfg, 'test')
        from .pointnet2 import PointNetSeg
        self.model = PointNetSeg(n_in=self.cfg['input_channel'], n_out=3 * self.cfg['ce_loss_bins'])
        self.model.load_state_dict(sd)
        self.model.to(self.device).eval()
        self._W = folding.prepare_seg(sd, self.device)
        self._mean, self._inv_std = transforms.normalizer_device(self.cfg, self.device)
        # With the package's own alignment under the reference's sampling, predict() draws the 2 x 10,000 hypothesis samples of
        # predicter.py:167-170 from numpy's stream on the worker thread WHILE the network runs (they depend on n_pts alone).
        self._predraw = align_fn is None and ransac_sampling == 'reference'
        if align_fn is None:          # estimate9DTransform (aligning.py:83-119): device RANSAC (catgrasp_amd/aligning.py, row N1)
            import functools
            from . import aligning
            align_fn = functools.partial(aligning.estimate9DTransform, sampling=ransac_sampling)
        self.align_fn = align_fn

    def nocs_on_device(self, cloud_xyz, cloud_normal, ids):
        """cloud (M,3) f32 cuda, ids (B,n_pts) i32 cuda -> coords (B,n_pts,3) in {k/bins-0.5}, conf_z (B,n_pts), logits."""
        x = ops.build_nunocs_input(cloud_xyz, cloud_normal, ids, self._mean, self._inv_std)
        logits = engine.run_guarded(engine.seg_forward, self._W, x)[0]
        B, N, _ = logits.shape
        nb = self.cfg['ce_loss_bins']
        coords, conf = ops.nunocs_decode(logits.view(B * N, 3 * nb), nb)
        return coords.view(B, N, 3), conf.view(B, N), logits

    def predict_nocs(self, data, ids=None):
        """The network + decode part of predict (predicter.py:135-150): returns (nocs_cloud (n_pts,3) float32,
        confidence_z (n_pts,), data_transformed dict with 'cloud_xyz_original', 'keep_ids')."""
        with torch.no_grad():
            cloud = transforms.DeviceCloud(data['cloud_xyz'], data['cloud_normal'], self.device)
            if ids is None:
                ids = transforms.draw_ids_reference(cloud.n, self.cfg['n_pts'], 1)[0]
            ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(1, -1)
            if ids.size and (ids.min() < 0 or ids.max() >= cloud.n):         # the device gather does no bounds checking
                raise IndexError(f'resample index out of range for a cloud of {cloud.n} valid points')
            hook, self._after_draw = getattr(self, '_after_draw', None), None
            if hook is not None:              # numpy's stream now stands where the reference's first estimate9DTransform finds it
                hook(ids.shape[1])
            coords, conf, _ = self.nocs_on_device(cloud.xyz, cloud.normal, torch.from_numpy(ids).to(self.device))
            self.data_transformed = {'cloud_xyz_original': cloud.xyz64[ids[0]].copy(), 'keep_ids': cloud.keep_ids[ids[0]],
                                     'cloud_normal': cloud.normal64[ids[0]].copy()}
            conf_h = conf[0].cpu().numpy()
            if not np.isfinite(conf_h).all():
                raise FloatingPointError('NUNOCS confidences are not finite (non-finite weights or activations beyond float32)')
            return coords[0].cpu().numpy(), conf_h, self.data_transformed

    RANSAC_THRESHOLDS, RANSAC_MAX_ITER = (0.003, 0.005), 10000             # predicter.py:167,170

    def draw_ahead(self, n_valid, state, pool):
        """The numpy-stream draws of ONE predict() call, made ahead of time from an explicit generator state on `pool`'s thread (a
        callable may be given for `state`: it is evaluated on that thread first -- e.g. transforms.advance_choice_rows over the
        scoring draws still in flight): the resampling row of the NUNOCS transform (dataset_nunocs.py:45-52) and the
        2 x 10,000 hypothesis samples (predicter.py:167-170 -> aligning.py:89-93).  -> a future of dict(start_state, ids, heads,
        end_state, seconds) for predict(..., predrawn=...); None when this predicter does not draw from numpy's stream that way."""
        n_pts = self.cfg['n_pts']
        if not self._predraw or n_pts < 4:
            return None

        def run():
            import time
            t0 = time.perf_counter()
            st0 = state() if callable(state) else state
            rows = transforms.NumpyChoiceStream(n_valid, n_pts, state=st0)
            ids = rows.draw(1)
            heads = transforms.NumpyHeadsDraw(n_pts, 4, len(self.RANSAC_THRESHOLDS) * self.RANSAC_MAX_ITER, state=rows.state())
            h = heads.result(set_state=False)
            return {'start_state': st0, 'n_valid': int(n_valid), 'ids': ids, 'after_ids_state': rows.state(), 'heads': h, 'end_state': heads.state(),
                    'seconds': time.perf_counter() - t0}
        return pool.submit(run)

    def predict(self, data, ids=None, predrawn=None, explicit_stream=False):
        """predicter.py:135-203: (nocs_cloud, 4x4 nocs_pose) or (None, None).  The 9-D RANSAC alignment
        (predicter.py:159-203 -> aligning.estimate9DTransform) runs on the device by default (`align_fn`).
        predrawn: the result of draw_ahead() for THIS call; used iff numpy's generator stands exactly where the draws started (and the
        cloud has the size they were made for) -- the call then returns the same values and leaves the same generator state as
        without it -- and silently ignored otherwise.
        explicit_stream (with predrawn): the draws were made from an explicit generator state and numpy's GLOBAL generator is neither
        checked nor moved -- pipeline.prepare_object running ahead of the global stream; the caller owns the stream position."""
        thresholds, max_iter = list(self.RANSAC_THRESHOLDS), self.RANSAC_MAX_ITER
        draw = []
        if explicit_stream and predrawn is None:
            raise ValueError('explicit_stream needs the pre-drawn values of draw_ahead()')
        if predrawn is not None and not explicit_stream and (
                ids is not None or not transforms.same_state(np.random.get_state(), predrawn['start_state'])
                or predrawn['n_valid'] != int(transforms.valid_mask(np.asarray(data['cloud_xyz'], dtype=np.float64)).sum())):
            predrawn = None
        if predrawn is not None:
            ids = predrawn['ids']

        def start_hypothesis_draw(n):
            if predrawn is None and self._predraw and n >= 4:
                draw.append(transforms.NumpyHeadsDraw(n, 4, len(thresholds) * max_iter, pool=_draw_worker()))
        import time
        t0 = time.perf_counter()
        self._after_draw = start_hypothesis_draw          # one-shot hook of predict_nocs (an overridden predict_nocs simply never calls it)
        try:
            nocs_cloud, _, dt = self.predict_nocs(data, ids)
        except BaseException as e:
            for d in draw:
                d.cancel()                     # the reference would not have reached its hypothesis draws either
            if predrawn is not None and not explicit_stream and isinstance(e, (FloatingPointError, IndexError)):
                # raised after the transform's resampling draw: the serial path has consumed that one row by now (draw_ids_reference in
                # predict_nocs), so the generator is left there whether or not the draws were made ahead
                np.random.set_state(predrawn['after_ids_state'])
            raise
        finally:
            self._after_draw = None
        t1 = time.perf_counter()
        hyp = draw[0].result() if draw else None
        if predrawn is not None:
            hyp = predrawn['heads']
            if not explicit_stream:
                np.random.set_state(predrawn['end_state'])      # where the reference's own draws would have left the generator
        t2 = time.perf_counter()
        ori = dt['cloud_xyz_original']
        best_ratio, best_transform = 0, None
        for k, thres in enumerate(thresholds):                              # predicter.py:167-198
            kw = {} if hyp is None else {'ids': hyp[k * max_iter:(k + 1) * max_iter]}
            transform, _ = self.align_fn(source=nocs_cloud.copy(), target=ori.copy(), PassThreshold=thres, max_iter=max_iter,
                                         use_kdtree_for_eval=False, kdtree_eval_resolution=0.003, max_scale=self.max_scale,
                                         min_scale=self.min_scale, max_dimensions=np.array([1.2, 1.2, 1.2]), **kw)
            if transform is None or np.linalg.det(transform[:3, :3]) < 0:
                continue
            transformed = (transform @ np.concatenate([nocs_cloud, np.ones((len(nocs_cloud), 1))], 1).T).T[:, :3]
            ratio = np.sum(np.linalg.norm(transformed - ori, axis=1) <= 0.003) / len(ori)
            if ratio > best_ratio:
                best_ratio, best_transform = ratio, transform.copy()
        # wall-clock split of this call (host clocks; the hypothesis draw runs on the worker thread under the network):
        # 'ransac id draw' = the C replay itself, 'ransac id draw (exposed)' = what the caller waited for it after the network
        self.timings = {'nunocs net + decode': t1 - t0, 'ransac id draw (exposed)': t2 - t1, 'ransac kernels + selection': time.perf_counter() - t2}
        if draw:
            self.timings['ransac id draw'] = draw[0].seconds
        if predrawn is not None:
            self.timings['ransac id draw'] = predrawn['seconds']          # spent earlier, on the draw-ahead thread
        if best_transform is None:
            return None, None
        self.best_ratio = best_ratio
        self.nocs_pose = best_transform.copy()
        return nocs_cloud, best_transform
