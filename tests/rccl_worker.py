"""One rank of the RCCL self-test (launched by tests/test_distributed_rccl_gpu.py, one process per GPU; RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the environment, as torch.distributed.run sets them).

Every rank builds the same small scenes with the REAL kernels, evaluates the whole batch itself (the unsharded reference), then runs
the sharded step -- its own slice + the ONE all_gather_into_tensor over RCCL (catgrasp_amd/distributed.py) -- and requires the gathered
records to equal the unsharded ones bit for bit.  With WORLD_SIZE=1 the collective is forced (`force_collective`), so a single-GPU box
still drives pad / all_gather_into_tensor / trim through RCCL.  Prints one JSON line per rank; exit code 0 = all equal."""
import json
import os
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                        # noqa: E402
import torch.distributed as dist    # noqa: E402


def main():
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from catgrasp_amd import distributed as cgd
    from catgrasp_amd import engine, synth, workload
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
    report = {'rank': rank, 'world': world, 'backend': dist.get_backend(), 'device': torch.cuda.get_device_name(local), 'cases': []}
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == world
        t = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(t)                                  # RCCL saw `world` ranks
        assert t.tolist() == [world * (world + 1) / 2.0] * 4, t.tolist()
        cats = ['nut', 'hnm', 'screw']
        sds = {c: (synth.make_state_dict('cls', 6, 10, seed=2 * i), synth.make_state_dict('seg', 6, 300, seed=2 * i + 1)) for i, c in enumerate(cats)}
        gps = {c: GraspPredicter(c, cfg=DEFAULT_GRASP_CFG, state_dict=sds[c][0], device=dev) for c in cats}
        nps = {c: NunocsPredicter(c, cfg=DEFAULT_NUNOCS_CFG, state_dict=sds[c][1], device=dev) for c in cats}
        # (name, kind, objects, evaluations per replica, replicas, arithmetic): C3's weak layout (slice r == replica r), C3 / C4 / C5 strong cuts
        # through symmetry groups and objects; odd totals so the padded tail of the gather is exercised.
        cases = [('C3-weak', 'nut', 3, 301, world, 'f32'), ('C3-strong', 'nut', 3, 1001, 1, 'f32'), ('C4-strong', 'screw', 4, 1203, 1, 'f32'),
                 ('C5-strong', 'bin', 4, 1501, 1, 'bf16x3'), ('tiny', 'nut', 1, 1, 1, 'f32')]
        for name, kind, n_obj, per, reps, prec in cases:
            engine.set_precision(prec)
            b = workload.SceneBatch(dev, gps, nps, kind=kind, n_objects=n_obj, pts_per_object=2100, per_replica=per, replicas=reps)
            with torch.no_grad():
                whole = b.score_slice(0, b.n_total)
                out = cgd.score_sharded(b.score_slice, b.n_total, force_collective=True)
            torch.cuda.synchronize()
            ok = out.shape == whole.shape and bool(torch.equal(out, whole))
            report['cases'].append({'case': name, 'n_total': b.n_total, 'precision': prec, 'equal': ok,
                                    'codes': torch.bincount(whole[:, 1].long(), minlength=5).tolist()})
            assert ok, f'{name}: gathered records differ from the unsharded ones on rank {rank}'
        report['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        report['ok'] = True
        dist.barrier()
    finally:                # a failing rank leaves without the barrier: its peers are reaped by the launcher's timeout
        print(json.dumps(report), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
