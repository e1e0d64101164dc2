"""CPU test (no GPU) of the multi-GPU path: world_size-2 `gloo` processes shard the candidates, score their
slice with a deterministic stand-in scorer, and the single all_gather reassembles the unsharded result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from catgrasp_amd import distributed as cgd


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_scores(lo, hi):
    i = torch.arange(lo, hi, dtype=torch.float32)
    return torch.stack([torch.sin(i) * 0.5 + 0.5, (i % 5)], dim=1)       # (p_G, code)


def _worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        out = cgd.score_sharded(_fake_scores, n_total)
        q.put((rank, out.clone()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(n_total, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_equals_unsharded_world2():
    for n_total in (1001, 8, 1):            # uneven tail, tiny, fewer candidates than ranks
        res = _run(n_total)
        ref = _fake_scores(0, n_total)
        for r in (0, 1):
            assert res[r].shape == ref.shape and torch.equal(res[r], ref)


def test_single_process_path():
    out = cgd.score_sharded(_fake_scores, 17)
    assert torch.equal(out, _fake_scores(0, 17))
