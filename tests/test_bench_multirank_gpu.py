"""GPU: first-contact insurance for the driver's multi-GPU run.  bench.py is launched exactly as the driver launches it for N = 8
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 ...`), with all 8 ranks on the ONE device of
this box and gloo as the process-group backend (CATGRASP_BENCH_BACKEND / CATGRASP_BENCH_DEVICE: dev switches of bench.py), on a small
batch: the emitted line must carry the contract's fields, and the gathered (p_G, code) records of the 8-rank job must be bit-identical
to those of the 1-rank job on the same batch (records_sha256)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--steps', '1', '--warmup', '1', '--candidates', '6000', '--secondary', '', '--no-cpu-baseline', '--no-api', '--no-pmc-traffic',
          '--no-rccl-selftest', '--no-projection', '--no-configs']


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                 # stdout carries exactly ONE json line
    return json.loads(lines[0])


def test_bench_line_of_an_8_rank_job_equals_the_1_rank_job(cuda_device):
    env = dict(os.environ, CATGRASP_BENCH_BACKEND='gloo', CATGRASP_BENCH_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    one = _run([sys.executable, 'bench.py', '--gpus', '1'] + COMMON, env)
    eight = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                  '--master-port', str(_free_port()), 'bench.py', '--gpus', '8'] + COMMON, env)
    for line, n in ((one, 1), (eight, 8)):
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                    'data', 'config', 'roofline', 'per_rank_ms', 'records_sha256'):
            assert key in line, key
        assert line['n_gpus'] == n and line['steps'] == 1 and line['scaling'] == 'strong' and line['higher_is_better'] is True
        assert line['config']['candidates_total'] == 6000 and line['value'] > 0 and 'workload' in line['config']
        assert set(line['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert len(eight['per_rank_ms']['ranks']) == 8 and eight['config']['candidates_per_gpu'] == 750
    # the census block of an N > 1 line (all 8 ranks sit on this box's one device here; on the driver's node they are 8 devices)
    assert eight['rccl']['ranks_seen'] == 8 and eight['rccl']['world_size'] == 8 and len(eight['rccl']['ranks']) == 8 and 'rccl' not in one
    assert eight['records_sha256'] == one['records_sha256']              # 8 shards + one all_gather == the unsharded batch, bit for bit
    assert eight['config']['reject_code_histogram_0keep_1dir_2ik_3open_4enclosed'] == one['config']['reject_code_histogram_0keep_1dir_2ik_3open_4enclosed']


@pytest.mark.parametrize('workload', ['C4', 'C5'])
def test_8_rank_job_of_the_8_gpu_configurations_equals_the_1_rank_job(cuda_device, workload):
    """the layouts the C3 test never takes -- C4 (configs[3]: 40k-point scene, 16 objects, hnm / screw symmetry counts) and
    C5 (configs[4]: mixed bin, 24 objects, three categories with their own predicters, bf16x3) -- as 8-rank jobs launched like the
    driver launches them, all ranks on this one device under gloo: the line carries the contract's fields and the gathered records
    are those of the 1-rank job."""
    env = dict(os.environ, CATGRASP_BENCH_BACKEND='gloo', CATGRASP_BENCH_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    common = ['--workload', workload, '--candidates-total', '4800', '--steps', '1', '--warmup', '1', '--secondary', '', '--no-cpu-baseline', '--no-api',
              '--no-pmc-traffic', '--no-rccl-selftest', '--no-projection', '--no-configs']
    one = _run([sys.executable, 'bench.py', '--gpus', '1'] + common, env)
    eight = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                  '--master-port', str(_free_port()), 'bench.py', '--gpus', '8'] + common, env)
    for line, n in ((one, 1), (eight, 8)):
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                    'data', 'config', 'roofline', 'per_rank_ms', 'records_sha256'):
            assert key in line, key
        assert line['n_gpus'] == n and line['scaling'] == 'strong' and line['config']['candidates_total'] == 4800 and line['value'] > 0
        assert line['config']['workload'].startswith(workload)
    assert len(eight['per_rank_ms']['ranks']) == 8 and eight['config']['candidates_per_gpu'] == 600
    assert eight['records_sha256'] == one['records_sha256']
    if workload == 'C5':
        assert 'bf16' in one['dtype'] and set(one['config']['symmetries']) == {'nut', 'hnm', 'screw'}
