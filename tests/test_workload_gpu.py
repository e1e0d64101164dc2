"""GPU tests of the shardable scene batch (catgrasp_amd/workload.py) that bench.py times: shard union == whole (bit for bit, for
slices that cut through symmetry groups and objects), records against the oracle on a sample, and the two-rank control flow."""
import numpy as np
import pytest
import torch

from catgrasp_amd import synth, workload
from oracle import collision_oracle as co
from oracle import pointnet_ref as oref
from oracle import transforms_ref as tref

pytestmark = pytest.mark.gpu
I4 = np.eye(4)


@pytest.fixture(scope='module')
def batch(cuda_device):
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    sd = synth.make_state_dict('cls', 6, 10, seed=0)
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=sd, device=cuda_device)
    npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=cuda_device)
    b = workload.SceneBatch(cuda_device, gp, npred, kind='nut', n_objects=3, pts_per_object=2200, per_replica=900, replicas=2)
    b.sd = sd
    return b


def test_shard_union_equals_whole(batch, mlp_precision):
    n = batch.n_total
    assert n == 1800 and sum(s.count for s in batch.segs) == n
    with torch.no_grad():
        whole = batch.score_slice(0, n)
        for cuts in ([0, 900, n], [0, 7, 150, 151, 449, 1000, 1777, n], [0, 5, n]):      # through symmetry groups, objects, replicas
            parts = [batch.score_slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
            assert torch.equal(torch.cat(parts), whole)
    assert whole.shape == (n, 2) and set(whole[:, 1].long().tolist()) <= {0, 1, 3, 4}
    assert (whole[:, 0] >= 0).all() and (whole[:, 0] <= 0.9 + 1e-6).all()


def test_one_filter_launch_per_slice_equals_one_call_per_segment(batch, bin_batch):
    """filter_launch='multi' (the default since round 6: every filter call of a slice through cg_filter_grasp_pose_multi) against
    'per_segment' (one filterGraspPose call per object and call shape on side streams): identical records, whole batch and odd cuts,
    single category and mixed bin."""
    for b in (batch, bin_batch):
        n = b.n_total
        assert b.filter_launch == 'multi'
        with torch.no_grad():
            multi = [b.score_slice(lo, hi) for lo, hi in ((0, n), (3, 501), (n // 2 - 1, n))]
            b.filter_launch = 'per_segment'
            try:
                per = [b.score_slice(lo, hi) for lo, hi in ((0, n), (3, 501), (n // 2 - 1, n))]
            finally:
                b.filter_launch = 'multi'
        for a, c in zip(multi, per):
            assert torch.equal(a, c)


def test_records_match_the_oracle_on_a_sample(batch, mlp_precision):
    """codes == the C oracle's for both call shapes (incl. nudged poses); p_G == the oracle network on the SAME resampled points
    (ids read back from the device draw, poses = the oracle's own output poses)."""
    from catgrasp_amd import transforms
    g = batch.gripper
    with torch.no_grad():
        rec = batch.score_slice(0, batch.n_total).cpu().numpy()
    for seg in batch.segs[:4]:
        P = batch.host_poses(seg)
        ob = batch.objs[seg.obj]
        bg = synth.background_points(batch.objs, seg.obj, g['diameter'])
        sym = transforms.get_symmetry_tfs('nut') if seg.kind == 'nocs' else [I4]
        nocs = batch.nocs_pose[seg.obj] if seg.kind == 'nocs' else I4
        oc, op, _ = co.filter_grasp_pose(P, sym, nocs, I4, I4, I4, g['gripper_in_grasp'], 1, 0, int(seg.adjust), g['vertices'], g['faces'],
                                         g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
        assert np.array_equal(rec[seg.start:seg.start + seg.count, 1].astype(np.int8), oc)
        keep = np.nonzero(oc == 0)[0][:6]
        if len(keep) == 0:
            continue
        dc = batch.clouds[seg.obj]
        xs = []
        for e in keep:
            ids = transforms.draw_ids_device(dc.n, 2048, 1, batch.device, seed=batch.draw_seed, row_offset=seg.start + int(e)).cpu().numpy()[0]
            xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), op[e].astype(np.float64), ids)['input'])
        logits = oref.pointnet_cls_forward(batch.sd, torch.from_numpy(np.stack(xs)).float())[0]
        probs = torch.softmax(logits, 1).numpy()
        pg = (probs * np.arange(10)).sum(1) / 10
        assert np.abs(rec[seg.start + keep, 0] - pg).max() <= 1e-4


@pytest.fixture(scope='module')
def bin_batch(cuda_device):
    """BASELINE.json configs[4] in small: nut + hnm + screw objects in one bin, one predicter pair per category (own weights)."""
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    cats = ['nut', 'hnm', 'screw']
    sds = {c: synth.make_state_dict('cls', 6, 10, seed=20 + i) for i, c in enumerate(cats)}
    gps = {c: GraspPredicter(c, cfg=DEFAULT_GRASP_CFG, state_dict=sds[c], device=cuda_device) for c in cats}
    nps = {c: NunocsPredicter(c, cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=30 + i), device=cuda_device)
           for i, c in enumerate(cats)}
    b = workload.SceneBatch(cuda_device, gps, nps, kind='bin', n_objects=4, pts_per_object=2100, per_replica=1500, replicas=1)
    b.sds = sds
    return b


def test_mixed_category_bin(bin_batch, mlp_precision):
    """Every object is filtered with its own category's symmetry set and scored under its own category's weights; the result does
    not depend on how the bin is cut (cuts through symmetry groups, objects and category changes)."""
    from catgrasp_amd import transforms
    b = bin_batch
    n = b.n_total
    assert n == 1500 and b.cats == ['nut', 'hnm', 'screw', 'nut']
    assert [s.n_sym for s in b.segs if s.kind == 'nocs'] == [12, 2, 72, 12]
    with torch.no_grad():
        whole = b.score_slice(0, n)
        for cuts in ([0, 375, 750, 1125, n], [0, 3, 380, 381, 760, 801, 1400, n]):
            assert torch.equal(torch.cat([b.score_slice(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]), whole)
    rec = whole.cpu().numpy()
    g = b.gripper
    for seg in b.segs:
        cat = b.cats[seg.obj]
        P = b.host_poses(seg)
        ob = b.objs[seg.obj]
        bg = synth.background_points(b.objs, seg.obj, g['diameter'])
        sym = transforms.get_symmetry_tfs(cat) if seg.kind == 'nocs' else [I4]
        nocs = b.nocs_pose[seg.obj] if seg.kind == 'nocs' else I4
        oc, op, _ = co.filter_grasp_pose(P, sym, nocs, I4, I4, I4, g['gripper_in_grasp'], 1, 0, int(seg.adjust), g['vertices'], g['faces'],
                                         g['enclosed_vertices'], g['enclosed_faces'], ob['xyz'], bg, 0.0005)
        assert np.array_equal(rec[seg.start:seg.start + seg.count, 1].astype(np.int8), oc)
        keep = np.nonzero(oc == 0)[0][:3]
        if len(keep) == 0:
            continue
        dc = b.clouds[seg.obj]
        xs = []
        for e in keep:
            ids = transforms.draw_ids_device(dc.n, 2048, 1, b.device, seed=b.draw_seed, row_offset=seg.start + int(e)).cpu().numpy()[0]
            xs.append(tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), op[e].astype(np.float64), ids)['input'])
        probs = torch.softmax(oref.pointnet_cls_forward(b.sds[cat], torch.from_numpy(np.stack(xs)).float())[0], 1).numpy()
        assert np.abs(rec[seg.start + keep, 0] - (probs * np.arange(10)).sum(1) / 10).max() <= 1e-4
