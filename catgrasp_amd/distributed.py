"""Multi-GPU candidate sharding: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Grasp candidates are independent given the read-only scene data (object clouds, weights, gripper meshes,
voxel sets -- all replicated, a few tens of MB), so the path shards by contiguous, equal slices of the
candidate array with NO data-path collective; the only exchange is ONE all_gather of a packed
per-candidate record (p_G float32, reject code) at the end.  The reference has no
distributed inference at all (nn.DataParallel is train-only, trainer_grasp.py:33)."""
import math

import torch
import torch.distributed as dist


def shard_bounds(n_total, world):
    """Equal contiguous slices of ceil(n/world); the tail slices may be short or empty."""
    per = int(math.ceil(n_total / max(world, 1))) if n_total > 0 else 0
    return per, [(min(n_total, r * per), min(n_total, (r + 1) * per)) for r in range(world)]


def gather_records(local_rec, per, n_total, group=None, force_collective=False):
    """local_rec: (n_local, C) tensor of this rank's slice (n_local <= per).  Pads to `per` rows (all_gather
    needs equal counts), gathers over all ranks in one collective and trims to (n_total, C), in candidate order.
    Device tensors go through ONE all_gather_into_tensor (RCCL over xGMI); under a gloo group (CPU tests, 1-GPU dev runs)
    the same buffers are exchanged through host memory.  A single rank skips the collective unless `force_collective` is set
    and a process group exists: the one-rank RCCL self-test (tests/test_distributed_rccl_gpu.py, bench.py `rccl_selftest`) drives
    the very same pad / all_gather_into_tensor / trim code the N-rank job runs."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local_rec[:n_total]
    C = local_rec.shape[1]
    buf = torch.zeros((per, C), dtype=local_rec.dtype, device=local_rec.device)
    buf[:local_rec.shape[0]] = local_rec
    if dist.get_backend(group) == 'nccl':
        out = torch.empty((world * per, C), dtype=local_rec.dtype, device=local_rec.device)
        dist.all_gather_into_tensor(out, buf, group=group)
    else:
        hbuf = buf.cpu()
        parts = [torch.empty_like(hbuf) for _ in range(world)]
        dist.all_gather(parts, hbuf, group=group)
        out = torch.cat(parts, 0).to(local_rec.device)
    return out[:n_total]


def score_sharded(score_fn, n_total, group=None, marks=None, force_collective=False):
    """Run score_fn(lo, hi) -> (hi-lo, C) on this rank's slice and return the full (n_total, C) record array
    on every rank.  score_fn sees global candidate indices.  `marks` (optional list): receives three device events per call
    -- start, after the local scoring, after the gather -- for per-phase timing."""
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    per, bounds = shard_bounds(n_total, world)
    lo, hi = bounds[rank]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if marks is not None else None
    if ev:
        ev[0].record()
    rec = score_fn(lo, hi)
    assert rec.shape[0] == hi - lo
    if ev:
        ev[1].record()
    out = gather_records(rec, per, n_total, group, force_collective)
    if ev:
        ev[2].record()
        marks.append(ev)
    return out
