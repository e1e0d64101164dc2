"""The N-rank half of tests/test_distributed_rccl_gpu.py: one process per GPU over RCCL with the REAL kernels (C3 weak / C3, C4, C5 strong
layouts in small); every rank requires gathered == unsharded bit for bit, and an all_reduce census proves RCCL saw N ranks.  Skipped on
a one-GPU box (where the one-rank test of that module drives the same collective code).  Named to be collected last."""
import pytest
import torch

from test_distributed_rccl_gpu import check, launch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs on the box (the 1-rank RCCL test always runs)')
def test_n_rank_rccl_all_gather_equals_unsharded(cuda_device):
    world = min(torch.cuda.device_count(), 8)
    check(launch(world), world)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs on the box')
def test_bench_py_itself_as_a_2_rank_rccl_job(cuda_device):
    """bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU, the real
    `nccl` backend = RCCL over xGMI) on a small C3 batch: the line's census shows 2 ranks on 2 distinct devices, and the gathered records
    are those of the 1-rank run bit for bit."""
    import os
    import sys
    from test_bench_multirank_gpu import COMMON, ROOT, _free_port, _run
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    env.pop('CATGRASP_BENCH_BACKEND', None); env.pop('CATGRASP_BENCH_DEVICE', None)
    one = _run([sys.executable, 'bench.py', '--gpus', '1'] + COMMON, env)
    two = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', str(_free_port()), 'bench.py', '--gpus', '2'] + COMMON, env)
    assert two['n_gpus'] == 2 and two['records_sha256'] == one['records_sha256']
    r = two['rccl']
    assert r['backend'] == 'nccl' and r['ranks_seen'] == 2 and r['distinct_devices'] == 2
    assert sorted(x['current_device'] for x in r['ranks']) == [0, 1] and sorted(x['rank'] for x in r['ranks']) == [0, 1]
    assert len(two['per_rank_ms']['ranks']) == 2
