"""CPU test (no GPU) of the host side of GraspPredicter.predict_batch: the chunk plan (ramp), the one-chunk-ahead worker that
replays numpy's stream, the protocol between score_on_device and its id source, list building -- with the device stages
(cg_apply_shuffle_rows, cg_build_grasp_input, the network, softmax) replaced at the tensor level by numpy/torch-CPU functions.
The product never runs like this (no CPU fallback): the mocks live in this test only."""
import numpy as np
import pytest
import torch

from catgrasp_amd import engine, ops, synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter


def _apply_shuffle_rows_host(partners, n_valid, n_pts, base=0, out=None):
    rows = []
    for js in partners.numpy():
        a = np.arange(n_valid)
        for s, j in enumerate(js[:n_valid - 1]):
            i = n_valid - 1 - s
            a[i], a[j] = a[j], a[i]
        rows.append(a[:n_pts] + base)
    return torch.from_numpy(np.stack(rows).astype(np.int32))


@pytest.fixture
def host_predicter(monkeypatch):
    seen = {'chunks': [], 'ids': []}

    def build_grasp_input(cloud_xyz, cloud_normal, ids, pose_inv, mean, inv_std):
        seen['chunks'].append(int(ids.shape[0])); seen['ids'].append(ids.clone())
        return ids                                        # the "network input" of a chunk is its id rows

    def cls_forward(W, x, status=None):
        f = x.double()
        return (torch.stack([f.sum(1) % 7, f[:, 0], f[:, -1], f.std(1)] + [f[:, k] * 1e-3 for k in range(6)], 1).float() * 1e-2, None)

    def softmax_pg(logits):
        p = torch.softmax(logits, 1)
        conf, label = p.max(1)
        return p, label.int(), conf, (p * torch.arange(10)).sum(1) / 10
    def pose_inverse_rows_f64(poses, center, bad=None):
        from catgrasp_amd import transforms
        return torch.from_numpy(transforms.pose_inverse_rows(poses.numpy().reshape(-1, 4, 4), center))

    class _Ev:
        def synchronize(self):
            pass
    from catgrasp_amd import predicter as pred_mod
    monkeypatch.setattr(pred_mod, '_pin', lambda shape, dtype: torch.empty(shape, dtype=dtype))
    monkeypatch.setattr(pred_mod, '_event', lambda: _Ev())
    monkeypatch.setattr(ops, 'pose_inverse_rows_f64', pose_inverse_rows_f64)
    monkeypatch.setattr(ops, 'apply_shuffle_rows', _apply_shuffle_rows_host)
    monkeypatch.setattr(ops, 'build_grasp_input', build_grasp_input)
    monkeypatch.setattr(engine, 'cls_forward', cls_forward)
    monkeypatch.setattr(ops, 'softmax_pg', softmax_pg)
    cfg = dict(DEFAULT_GRASP_CFG); cfg['n_pts'] = 16
    gp = GraspPredicter('nut', cfg=cfg, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device='cpu', chunk=64)
    return gp, seen


@pytest.mark.parametrize('n_cloud,G', [(40, 150), (16, 5), (12, 70), (40, 64), (40, 1)])
def test_numpy_mode_pipeline_equals_the_reference_loop(host_predicter, n_cloud, G):
    """rng='numpy' hands every pose the row np.random.choice would have drawn for it, in order, through a ramped chunk plan, and
    leaves numpy's generator where the reference's loop leaves it; n_cloud < n_pts takes the replace=True branch (host draw)."""
    gp, seen = host_predicter
    ob = synth.make_scene(1, n_cloud, seed=1)[0]
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    P = list(synth.make_candidates(ob, G, np.random.default_rng(2)))
    np.random.seed(3)
    want_ids = np.stack([np.random.choice(np.arange(n_cloud), size=(16), replace=n_cloud < 16) for _ in range(G)])
    after_want = np.random.rand()
    want = gp.predict_batch(data, P, ids=want_ids)
    explicit_chunks = list(seen['chunks']); seen['chunks'].clear(); seen['ids'].clear()
    np.random.seed(3)
    got = gp.predict_batch(data, P)                       # default: rng='numpy'
    after_got = np.random.rand()
    assert after_got == after_want
    assert np.array_equal(torch.cat(seen['ids']).numpy(), want_ids)
    assert len(got) == G and all(a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) for a, b in zip(got, want))
    assert isinstance(got[0][0], (int, np.integer)) and got[0][2].shape == (10,) and got[0][2].dtype == np.float32
    # chunk plans: the ramp (2048, 4096, 8192, then full chunks) capped by the predicter's chunk size -- 64 here, i.e. uniform
    assert explicit_chunks == [min(64, G - s) for s in range(0, G, 64)]
    assert sum(seen['chunks']) == G and all(c <= 64 for c in seen['chunks'])


def test_chunk_ramp_plan(host_predicter):
    """An id source with a ramp is asked for (2048, 4096, 8192, then full chunks) capped by the predicter's chunk size, each chunk
    exactly once and in order, and is told the plan before the first request."""
    gp, seen = host_predicter
    gp.chunk = 16384
    asked, planned = [], []

    def ids(s, e):
        asked.append((s, e))
        return torch.zeros((e - s, 16), dtype=torch.int32)
    ids.ramp = (2048, 4096, 8192)
    ids.plan = lambda bounds: planned.append(list(bounds))
    G = 50000
    gp.score_on_device(None, None, ids, torch.zeros((G, 12)))
    assert asked == [(0, 2048), (2048, 6144), (6144, 14336), (14336, 30720), (30720, 47104), (47104, 50000)]
    assert planned == [asked] and seen['chunks'] == [e - s for s, e in asked]
    gp.chunk = 3000; asked.clear(); planned.clear()
    gp.score_on_device(None, None, ids, torch.zeros((7000, 12)))
    assert asked == [(0, 2048), (2048, 5048), (5048, 7000)]


def test_predict_batch_chunk_plan_collector_pause_and_shared_worker(host_predicter):
    """The reference-API call: its chunk plan (1,024 growing by 1.5x up to the predicter's chunk size, every pose exactly once), the
    garbage collector paused for the call and restored to the caller's setting on return AND on an exception, one process-wide
    numpy-stream worker thread."""
    import gc
    from catgrasp_amd import predicter as pred_mod
    gp, _ = host_predicter
    gp.chunk = 16384
    b = gp._chunk_plan(50000, None)
    assert [e - s for s, e in b] == [1024, 1536, 2304, 3456, 5184, 7776, 11664, 16384, 672] and b[0][0] == 0 and b[-1][1] == 50000
    assert all(a[1] == c[0] for a, c in zip(b[:-1], b[1:]))
    assert gp._chunk_plan(700, None) == [(0, 700)] and gp._chunk_plan(1024, None) == [(0, 1024)] and gp._chunk_plan(1025, None) == [(0, 1024), (1024, 1025)]
    gp.chunk = 3000
    assert [e - s for s, e in gp._chunk_plan(9000, None)] == [1024, 1536, 2304, 3000, 1136]
    for was in (True, False):
        (gc.enable if was else gc.disable)()
        try:
            with pred_mod._gc_paused():
                assert not gc.isenabled()
            assert gc.isenabled() == was
            with pytest.raises(KeyError):
                with pred_mod._gc_paused():
                    raise KeyError('x')
            assert gc.isenabled() == was
        finally:
            gc.enable()
    assert pred_mod._draw_worker() is pred_mod._draw_worker()


def test_explicit_resample_ids_are_range_checked_before_any_device_work():
    """The device gathers do no bounds checking: explicit ids outside [0, n_valid) must raise like numpy indexing would in the
    reference, for both predicters, before anything is launched (so this runs without a GPU)."""
    from catgrasp_amd.predicter import DEFAULT_NUNOCS_CFG, NunocsPredicter
    ob = synth.make_scene(1, 300, seed=1)[0]
    data = {'cloud_xyz': ob['xyz'], 'cloud_normal': ob['normal']}
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device='cpu')
    n_pts = DEFAULT_NUNOCS_CFG['n_pts']
    for bad in (-1, 300):
        ids = np.zeros(n_pts, dtype=np.int64); ids[7] = bad
        with pytest.raises(IndexError):
            npred.predict_nocs(data, ids=ids)
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device='cpu')
    P = list(synth.make_candidates(ob, 2, np.random.default_rng(2)))
    ids = np.zeros((2, DEFAULT_GRASP_CFG['n_pts']), dtype=np.int64); ids[1, 3] = 300
    with pytest.raises(IndexError):
        gp.predict_batch(data, P, ids=ids)
    with pytest.raises(ValueError):
        gp.predict_batch(data, P, ids=ids[:1])
