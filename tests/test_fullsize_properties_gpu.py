"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot finish 10^4..10^5 candidates in
seconds): candidate-permutation equivariance, point-order invariance (PointNet symmetry), shard-union == whole,
reduction identities, batch-composition independence of the collision codes, broad phase == exhaustive -- all BIT-EXACT --
plus an oracle spot check of a random sample of the same batch.

Sizes: C2 = 20k-pt scene (8 x 2500), 10k candidates; C3 = same scene, 50k candidates through NUNOCS + grasp-Q + collision;
C4 = screw category, 40k-pt scene (16 x 2500), 200k candidates (the 8-GPU configuration, run here on one device in 8 shards);
C5 = mixed-category bin (nut + screw + screw per triple, 24 objects = 60k points), 500k candidates, split-bf16 MFMA path."""
import numpy as np
import pytest
import torch

from catgrasp_amd import distributed, my_cpp, synth, workload
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
from oracle import collision_oracle as co
from oracle import pointnet_ref as oref
from oracle import transforms_ref as tref

pytestmark = pytest.mark.gpu


_WORKLOADS = {}
_ORACLE_PROBS = {}
N_ORACLE = 1024        # oracle-scored candidates per configuration; computed once, shared by the arithmetic modes


def _oracle_probs(key, sd, make_inputs):
    """softmax of the oracle network (the torch-nn-ops formulation, pinned to the reference's golden outputs in
    tests/test_oracle_golden.py) on N_ORACLE candidates, in chunks of 200 like predicter.py:69; cached per configuration."""
    if key not in _ORACLE_PROBS:
        import os
        x = torch.from_numpy(make_inputs()).float()
        psd = oref.prepared_state_dict(sd)
        old = torch.get_num_threads()
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        try:
            with torch.no_grad():
                lg = torch.cat([oref.pointnet_cls_forward_nnops(psd, x[s:s + 200])[0] for s in range(0, len(x), 200)])
        finally:
            torch.set_num_threads(old)
        _ORACLE_PROBS[key] = torch.softmax(lg, 1).numpy()
    return _ORACLE_PROBS[key]


def _flat_workload(device, G, seed, n_objects, kind='nut'):
    """build_flat_workload, built once per configuration: the arithmetic modes of a test share the (read-only) synthetic batch --
    generating 10^5 candidate poses is a host-side python loop that would otherwise be repeated for every mode."""
    key = (str(device), G, seed, n_objects, kind)
    if key not in _WORKLOADS:
        _WORKLOADS[key] = workload.build_flat_workload(device, G, seed=seed, n_objects=n_objects, pts_per_object=2500, kind=kind)
    return _WORKLOADS[key]


def _candidate_owner(wl):
    """global candidate index -> (object k, local index j), following workload.build_flat_workload's per-object blocks"""
    start = np.concatenate([[0], np.cumsum(wl['per'])])
    return start


@pytest.mark.parametrize('n_obj,G,kind', [(8, 10000, 'nut'), (16, 200000, 'screw')])
def test_scoring_properties_at_baseline_sizes(cuda_device, n_obj, G, kind):
    wl = _flat_workload(cuda_device, G, 0, n_obj, kind)
    assert wl['cloud_xyz'].shape[0] == n_obj * 2500 and wl['ids'].shape == (G, 2048)
    sd = synth.make_state_dict('cls', 6, 10, seed=0)
    gp = GraspPredicter(kind, cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=cuda_device)
    xyz, nrm, ids, pinv = wl['cloud_xyz'], wl['cloud_normal'], wl['ids'], wl['pose_inv']
    with torch.no_grad():
        probs, label, conf, pg = gp.score_on_device(xyz, nrm, ids, pinv)
        # reduction identities (predicter.py:86-91, run_grasp_simulation.py:311-319)
        assert (probs.sum(1) - 1).abs().max().item() < 1e-5
        assert torch.equal(label.long(), probs.argmax(1)) and torch.equal(conf, probs.max(1).values)
        k = torch.arange(10, device=cuda_device, dtype=torch.float32)
        assert ((probs * k).sum(1) / 10 - pg).abs().max().item() < 1e-6
        assert torch.isfinite(probs).all()
        # candidate-permutation equivariance: a candidate's score does not depend on where it sits in the batch
        gen = torch.Generator(device='cpu'); gen.manual_seed(5)
        perm = torch.randperm(G, generator=gen).to(cuda_device)
        probs_p = gp.score_on_device(xyz, nrm, ids[perm].contiguous(), pinv[perm].contiguous())[0]
        assert torch.equal(probs_p, probs[perm])
        # point-order invariance: the max-pool makes the network a set function of the 2048 resampled points
        probs_r = gp.score_on_device(xyz, nrm, ids.flip(1).contiguous(), pinv)[0]
        assert torch.equal(probs_r, probs)
        # shard union == whole (the multi-GPU partition of SURVEY 8(e), 8 ranks, evaluated on one device)
        per, bounds = distributed.shard_bounds(G, 8)
        parts = [gp.score_on_device(xyz, nrm, ids[lo:hi], pinv[lo:hi])[0] for lo, hi in bounds if hi > lo]
        assert torch.equal(torch.cat(parts), probs)
    # the oracle on 1,024 candidates of the same batch, spread over every object: transform restatement + oracle network
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(G, N_ORACLE, replace=False))
    start = _candidate_owner(wl)
    owners = np.searchsorted(start, pick, side='right') - 1
    assert set(owners.tolist()) == set(range(n_obj))

    def make_inputs():
        ids_h = ids[torch.from_numpy(pick).to(cuda_device)].cpu().numpy()
        poses_h = [p.cpu().numpy().reshape(-1, 4, 4).astype(np.float64) for p in wl['poses_dev']]
        xs = []
        for g, row, kob in zip(pick, ids_h, owners):
            ob = wl['objs'][kob]
            xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), poses_h[kob][g - start[kob]], row - kob * 2500)['input'])
        return np.stack(xs)
    ref_probs = _oracle_probs((n_obj, G, kind), sd, make_inputs)
    got = probs[torch.from_numpy(pick).to(cuda_device)].cpu().numpy()
    assert np.abs(got - ref_probs).max() <= 1e-4
    ref_pg = (ref_probs * np.arange(10)).sum(1) / 10
    assert np.abs(pg[torch.from_numpy(pick).to(cuda_device)].cpu().numpy() - ref_pg).max() <= 1e-4


def test_mixed_category_bin_at_c5_size(cuda_device, mlp_precision):
    """configs[4]: every object is scored by its own category's predicter (run_grasp_simulation.py keeps one GraspPredicter per
    class); shard-union == whole per category and an oracle spot check per category."""
    if mlp_precision != 'bf16x3':
        pytest.skip('configs[4] names the bf16 MFMA path')
    G = 500000
    wl = workload.build_flat_workload(cuda_device, G, seed=2, n_objects=24, pts_per_object=2500, kind='mixed')
    kinds = [ob['kind'] for ob in wl['objs']]
    assert set(kinds) == {'nut', 'screw'} and wl['cloud_xyz'].shape[0] == 60000
    start = _candidate_owner(wl)
    owner = torch.from_numpy(np.repeat(np.arange(24), wl['per'])).to(cuda_device)
    is_nut = torch.tensor([k == 'nut' for k in kinds], device=cuda_device)[owner]
    sds = {'nut': synth.make_state_dict('cls', 6, 10, seed=40), 'screw': synth.make_state_dict('cls', 6, 10, seed=41)}
    rng = np.random.default_rng(9)
    scored = 0
    with torch.no_grad():
        for cat in ('nut', 'screw'):
            gp = GraspPredicter(cat, cfg=DEFAULT_GRASP_CFG, state_dict=sds[cat], device=cuda_device)
            sel = torch.nonzero(is_nut if cat == 'nut' else ~is_nut)[:, 0]
            ids, pinv = wl['ids'][sel].contiguous(), wl['pose_inv'][sel].contiguous()
            probs = gp.score_on_device(wl['cloud_xyz'], wl['cloud_normal'], ids, pinv)[0]
            scored += probs.shape[0]
            per, bounds = distributed.shard_bounds(len(sel), 8)
            parts = [gp.score_on_device(wl['cloud_xyz'], wl['cloud_normal'], ids[lo:hi], pinv[lo:hi])[0] for lo, hi in bounds if hi > lo]
            assert torch.equal(torch.cat(parts), probs)
            pick = np.sort(rng.choice(len(sel), 16, replace=False))
            xs = []
            for j in pick:
                g = int(sel[j]); kob = int(np.searchsorted(start, g, side='right') - 1)
                ob = wl['objs'][kob]
                P = wl['poses_dev'][kob][g - start[kob]].cpu().numpy().reshape(4, 4).astype(np.float64)
                xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P, ids[j].cpu().numpy() - kob * 2500)['input'])
            ref = torch.softmax(oref.pointnet_cls_forward(sds[cat], torch.from_numpy(np.stack(xs)).float())[0], 1).numpy()
            assert np.abs(probs[torch.from_numpy(pick).to(cuda_device)].cpu().numpy() - ref).max() <= 1e-4
    assert scored == G


def test_collision_and_nunocs_properties_at_c3_size(cuda_device):
    G = 50000
    wl = workload.build_flat_workload(cuda_device, G, seed=1, n_objects=8, pts_per_object=2500)
    g = wl['gripper']
    I4 = np.eye(4)
    sym_h = np.stack([nut_symmetry(i) for i in range(12)])      # nut: 12 symmetry transforms (Utils.py:79-94)
    sym = torch.from_numpy(sym_h.astype(np.float32)).to(cuda_device)
    total = n_oracle = 0
    for kob in range(8):                                          # every object of the scene
        ob, sc = wl['objs'][kob], wl['scenes'][kob]
        # grasp_sampler.py:345 call shape: canonical-frame grasps, 9-D nocs_pose (anisotropic scale), symmetry expansion, nudging
        nocs_pose = ob['pose'] @ np.diag([0.016, 0.016, 0.006, 1.0])
        P_cam = wl['poses_dev'][kob].cpu().numpy().reshape(-1, 4, 4).astype(np.float64)
        P_can = np.linalg.inv(nocs_pose) @ P_cam
        P = torch.from_numpy(P_can.astype(np.float32).reshape(-1, 16)).to(cuda_device)
        n = P.shape[0]
        args = (nocs_pose, I4, I4, I4, g['gripper_in_grasp'], False, False, True)
        codes, poses, nudge = my_cpp.filter_on_device(sc, P, sym, *args)
        E = n * 12
        total += E
        hist = np.bincount(codes.cpu().numpy(), minlength=5)
        assert codes.shape == (E,) and hist[0] > 0 and hist[3] + hist[4] > 0 and hist[1] == hist[2] == 0, hist
        # batch-composition independence, bit-exact: a random subset filtered alone gives the same codes / poses / nudges
        rng = np.random.default_rng(10 + kob)
        sub = np.sort(rng.choice(n, 100, replace=False))
        sub_t = torch.from_numpy(sub).to(cuda_device)
        c2, p2, n2 = my_cpp.filter_on_device(sc, P[sub_t].contiguous(), sym, *args)
        e_idx = (sub_t[:, None] * 12 + torch.arange(12, device=cuda_device)[None]).reshape(-1)
        assert torch.equal(c2, codes[e_idx]) and torch.equal(n2, nudge[e_idx])
        keep = c2 == 0
        assert torch.equal(p2[keep], poses[e_idx][keep])
        bg = synth.background_points(wl['objs'], kob, g['diameter'])
        if kob in (0, 5):     # broad phase == exhaustive kernel at full size
            sc_ex = my_cpp.GripperScene(g['vertices'], g['faces'], g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005,
                                        cuda_device, accel=False)
            c3, p3, n3 = my_cpp.filter_on_device(sc_ex, P, sym, *args)
            assert torch.equal(c3, codes) and torch.equal(n3, nudge) and torch.equal(p3[codes == 0], poses[codes == 0])
        # the C oracle on the subset (bit-exact codes and nudge indices, poses of the survivors): 1,200 evaluations per object
        oc, op, on = co.filter_grasp_pose(P[sub_t].cpu().numpy().reshape(-1, 4, 4), list(sym_h), nocs_pose, I4, I4, I4,
                                          g['gripper_in_grasp'], 0, 0, 1, g['vertices'], g['faces'], g['enclosed_vertices'],
                                          g['enclosed_faces'], ob['xyz'], bg, 0.0005)
        assert np.array_equal(oc, c2.cpu().numpy()) and np.array_equal(on, n2.cpu().numpy())
        assert np.array_equal(op[oc == 0], p2.cpu().numpy()[oc == 0])
        n_oracle += len(oc)
        # grasp_sampler.py:216 call shape on the same object: camera-frame poses, symmetry [I], approach-direction filter, no nudging
        Pc = torch.from_numpy(P_cam.astype(np.float32).reshape(-1, 16)).to(cuda_device)
        eye = torch.eye(4, device=cuda_device).reshape(1, 16)
        cc, pc, _ = my_cpp.filter_on_device(sc, Pc, eye, I4, I4, I4, I4, g['gripper_in_grasp'], True, False, False)
        sub2 = np.sort(rng.choice(n, 150, replace=False))
        oc2, op2, _ = co.filter_grasp_pose(P_cam[sub2].astype(np.float32), [I4], I4, I4, I4, I4, g['gripper_in_grasp'], 1, 0, 0, g['vertices'],
                                           g['faces'], g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
        sub2_t = torch.from_numpy(sub2).to(cuda_device)
        assert np.array_equal(oc2, cc[sub2_t].cpu().numpy()) and np.array_equal(op2[oc2 == 0], pc[sub2_t].cpu().numpy()[oc2 == 0])
        n_oracle += len(oc2)
    assert n_oracle >= 10000
    assert total == 8 * 6250 * 12
    # NUNOCS over the 8 object clouds of the scene: every decoded coordinate is a bin centre in [-0.5, 0.5), and the batch of 8
    # equals 8 single-cloud calls bit for bit
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=cuda_device)
    with torch.no_grad():
        coords, conf, _ = npred.nocs_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['nunocs_ids'])
        assert coords.shape == (8, 8192, 3)
        binned = (coords + 0.5) * 100
        assert (binned - binned.round()).abs().max().item() < 1e-4 and coords.min().item() >= -0.5 and coords.max().item() < 0.5
        for k in (0, 7):
            ck, fk, _ = npred.nocs_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['nunocs_ids'][k:k + 1].contiguous())
            assert torch.equal(ck[0], coords[k]) and torch.equal(fk[0], conf[k])
        # ... and against the oracle at full size: the logits of ALL 8 x 8192 points from the oracle network on the restated
        # NunocsIsolatedDataset.transform of the same resampled points; decoded coordinates equal wherever the top-2 bin logits are
        # separated by more than the tolerance
        _, _, logits = npred.nocs_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['nunocs_ids'])
    import os
    ids_h = wl['nunocs_ids'].cpu().numpy()
    xin = np.stack([tref.nunocs_transform(wl['objs'][k]['xyz'].copy(), wl['objs'][k]['normal'].copy(), ids_h[k] - k * 2500)['input'] for k in range(8)])
    sd_seg = synth.make_state_dict('seg', 6, 300, seed=1)
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        with torch.no_grad():
            ref_lg = torch.cat([oref.pointnet_seg_forward_nnops(oref.prepared_state_dict(sd_seg), torch.from_numpy(xin[k:k + 1]).float())[0] for k in range(8)])
    finally:
        torch.set_num_threads(old)
    ref_lg = ref_lg.numpy()
    got_lg = logits.cpu().numpy()
    assert got_lg.shape == ref_lg.shape == (8, 8192, 300)
    assert (np.abs(got_lg - ref_lg) / np.maximum(1.0, np.abs(ref_lg))).max() <= 1e-4
    srt = np.sort(ref_lg.reshape(8, 8192, 3, 100), axis=-1)
    clear = (srt[..., -1] - srt[..., -2]) > 2e-4
    ref_coords = tref.nunocs_decode(ref_lg.reshape(-1, 300), 100)[0].reshape(8, 8192, 3)      # predicter.py:144-150, float32 like the reference
    assert clear.mean() > 0.99 and np.array_equal(coords.cpu().numpy()[clear], ref_coords[clear])


def nut_symmetry(i):
    """nut symmetry set: 6 rotations about z (60 degree steps) x flip about x (Utils.py:79-94 builds the same group)"""
    a = (i % 6) * np.pi / 3
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    if i >= 6:
        R = R @ np.diag([1.0, -1.0, -1.0])
    T = np.eye(4)
    T[:3, :3] = R
    return T
