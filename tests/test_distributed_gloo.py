"""CPU tests (no GPU) of the multi-GPU path: world_size-2 `gloo` processes run the REAL step control flow of bench.py
(catgrasp_amd/workload.py: segment plan, slice intersection, rectangle splitting, per-object prep runs, record packing;
catgrasp_amd/distributed.py: shard bounds, padding, the single all_gather, trimming) with the four device stages replaced at the
tensor level by deterministic functions of the GLOBAL evaluation index -- so the gathered result must equal the unsharded one
exactly, for weak-scaling layouts (slice == replica) and strong-scaling cuts that run through symmetry groups and objects."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from catgrasp_amd import distributed as cgd
from catgrasp_amd import synth, workload


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


class HostBatch(workload.SceneBatch):
    """SceneBatch with its device stages mocked on CPU tensors; everything else is the product's code."""

    def __init__(self, n_objects, per_replica, n_sym, replicas):
        self.device = torch.device('cpu')
        if isinstance(n_sym, str):      # a mixed-category bin of synth.MIXED_BINS: per-object category, symmetry count and weights
            self.cats = [synth.MIXED_BINS[n_sym][k % 3] for k in range(n_objects)]
        else:
            self.cats = [{12: 'nut', 2: 'hnm', 72: 'screw'}[n_sym]] * n_objects
        per_obj = [workload.SYMMETRY_COUNT[c] for c in self.cats]
        self.n_sym = max(per_obj)
        self.segs, self.n_total = workload.plan_segments(n_objects, per_replica, per_obj, replicas)
        self.nunocs_calls, self.net_calls = [], []

    def run_nunocs(self, obj_ids):
        self.nunocs_calls.append(list(obj_ids))

    def run_filter(self, seg, i0, i1, j0, j1):
        i = torch.arange(i0, i1).view(-1, 1); j = torch.arange(j0, j1).view(1, -1)
        e = (seg.start + i * seg.n_sym + j).reshape(-1)                     # evaluation order of a (poses x symmetries) launch
        codes = ((e * 7 + seg.obj) % 5).to(torch.int8)
        poses = (e.view(-1, 1) * 16 + torch.arange(16).view(1, -1)).float() * (1.0 if seg.adjust else -1.0)
        return codes, poses

    def run_filter_many(self, key, rects):
        out = [self.run_filter(*r) for r in rects]
        return torch.cat([c for c, _ in out]), torch.cat([p for _, p in out])

    def alloc(self, n):
        return torch.empty((n, 12)), torch.empty((n, 5), dtype=torch.int32)

    def run_prep(self, obj, poses, row_offset, pinv_out, ids_out):
        pinv_out.copy_(poses[:, :12] + 1000.0 * obj)
        ids_out[:, :4] = (row_offset + torch.arange(poses.shape[0])).view(-1, 1).to(torch.int32) * 4 + torch.arange(4, dtype=torch.int32)
        ids_out[:, 4] = obj

    def run_net(self, ids, pinv, cat=None):
        # every row must arrive under the weights of its own object's category (the object id rides in ids[:, 4], see run_prep)
        assert all(self.cats[k] == cat for k in ids[:, 4].unique().tolist())
        self.net_calls.append((cat, int(ids.shape[0])))
        w = {'nut': 1.0, 'hnm': 2.0, 'screw': 3.0}[cat]
        return torch.sin(ids[:, 1].float() * 0.37) * 0.25 * w + pinv[:, 3] * 1e-6


def _worker(rank, world, port, cfg, q, force=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        b = HostBatch(*cfg)
        out = cgd.score_sharded(b.score_slice, b.n_total, force_collective=force)
        q.put((rank, out.numpy().copy(), b.nunocs_calls))      # by value: a shared-memory tensor could outlive its producer
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(cfg, world=2, force=False):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, q, force)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return {r: (torch.from_numpy(o), c) for r, o, c in got}


def test_sharded_step_equals_unsharded_world2():
    for cfg in ((3, 101, 12, 2),        # weak layout: 2 replicas x 101 candidates, slice r == replica r
                (4, 1001, 12, 1),       # strong: odd total, the cut falls inside a symmetry group
                (16, 2000, 72, 1),      # C4 shape in small: 16 objects, 72 symmetries
                (24, 5003, 'bin', 1),   # C5 shape in small: mixed nut + hnm + screw bin (12 / 2 / 72 symmetries, per-category weights)
                (5, 600, 'bin', 2),     # the same bin under the weak layout
                (2, 3, 12, 1), (1, 1, 12, 1)):     # fewer candidates than ranks
        whole = HostBatch(*cfg)
        ref = whole.score_slice(0, whole.n_total)
        assert ref.shape == (whole.n_total, 2)
        res = _run(cfg)
        for r in (0, 1):
            assert res[r][0].shape == ref.shape and torch.equal(res[r][0], ref), cfg
        if cfg[3] == 2:                 # weak scaling: every rank ran the NUNOCS stage over all objects of its replica
            assert res[0][1] == [list(range(cfg[0]))] and res[1][1] == [list(range(cfg[0]))]
        if cfg[2] == 'bin':             # one scoring batch per run of same-category rows, all categories present
            assert {c for c, _ in whole.net_calls} == {'nut', 'hnm', 'screw'}
            assert sum(n for _, n in whole.net_calls) == whole.n_total and len(whole.net_calls) == cfg[0] * cfg[3]


def test_sharded_step_equals_unsharded_at_the_world_sizes_the_driver_launches():
    """The 4- and 8-rank layouts of `bench.py --gpus N` (strong C3 / C4 / C5 shapes in small, the weak layout, a batch with fewer
    candidates than ranks): uneven tails, empty shards, pad -> all_gather -> trim over more than two ranks, every rank holding
    the whole result."""
    for world, cfgs in ((4, ((8, 1003, 12, 1), (3, 50, 12, 4))), (8, ((16, 2001, 72, 1), (24, 1507, 'bin', 1), (2, 5, 12, 1)))):
        for cfg in cfgs:
            whole = HostBatch(*cfg)
            ref = whole.score_slice(0, whole.n_total)
            res = _run(cfg, world=world)
            assert sorted(res) == list(range(world))
            for r in range(world):
                assert torch.equal(res[r][0], ref), (world, cfg, r)


def test_one_rank_group_with_the_collective_forced():
    """The shape of the 1-GPU RCCL self-test (tests/test_distributed_rccl_gpu.py, bench.py `rccl_selftest`): a 1-rank group whose
    gather is forced through the collective returns the unsharded records; without a process group the flag is inert."""
    cfg = (3, 101, 12, 1)
    whole = HostBatch(*cfg)
    ref = whole.score_slice(0, whole.n_total)
    assert torch.equal(_run(cfg, world=1, force=True)[0][0], ref)
    assert torch.equal(cgd.score_sharded(whole.score_slice, whole.n_total, force_collective=True), ref)


def test_slicing_arithmetic():
    segs, n = workload.plan_segments(8, 50000, 12, 1)
    assert n == 50000 and segs[0].count == 260 * 12 and segs[1].count == 3130 and all(s.count > 0 for s in segs)
    assert sum(s.count for s in segs) == n and all(a.start + a.count == b.start for a, b in zip(segs[:-1], segs[1:]))
    for n_sym, a, b in ((12, 5, 7), (12, 5, 40), (12, 12, 36), (1, 3, 9), (12, 0, 5), (72, 71, 73), (12, 0, 0)):
        rects = workload.split_eval_range(n_sym, a, b)
        ev = [i * n_sym + j for i0, i1, j0, j1 in rects for i in range(i0, i1) for j in range(j0, j1)]
        assert ev == list(range(a, b)) and len(rects) <= 3
    per, bounds = cgd.shard_bounds(200000, 8)
    assert per == 25000 and bounds[7] == (175000, 200000)
    assert cgd.shard_bounds(10, 4)[1] == [(0, 3), (3, 6), (6, 9), (9, 10)] and cgd.shard_bounds(2, 4)[1][3] == (2, 2)


def test_mixed_bin_plan():
    """Per-object symmetry counts: each object's 'nocs' segment uses its own category's count; totals are exact."""
    cats = [synth.MIXED_BINS['bin'][k % 3] for k in range(24)]
    segs, n = workload.plan_segments(24, 500000, [workload.SYMMETRY_COUNT[c] for c in cats], 1)
    assert n == 500000 and sum(s.count for s in segs) == n
    for s in segs:
        assert s.n_sym == (workload.SYMMETRY_COUNT[cats[s.obj]] if s.kind == 'nocs' else 1)
    assert all(a.start + a.count == b.start for a, b in zip(segs[:-1], segs[1:]))
    per_obj = {}
    for s in segs:
        per_obj[s.obj] = per_obj.get(s.obj, 0) + s.count
    assert sorted(set(per_obj.values())) == [20833, 20834]


def test_single_process_path():
    b = HostBatch(3, 50, 12, 1)
    assert torch.equal(cgd.score_sharded(b.score_slice, b.n_total), b.score_slice(0, b.n_total))
