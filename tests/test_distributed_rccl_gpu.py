"""GPU tests of the ONE collective of the multi-GPU path, on RCCL itself (BASELINE.json north_star:
"a single RCCL all-gather of scores over xGMI").

* one rank (always runs on the 1-GPU box): a 1-rank `nccl` process group with the collective FORCED, so pad -> all_gather_into_tensor
  -> trim of catgrasp_amd/distributed.py run on RCCL with the real SceneBatch step (C3 weak / C3, C4, C5 strong layouts in small) and
  must return exactly the unsharded records;
* N ranks (tests/test_zzz_rccl_multirank_gpu.py -- collected LAST, so a multi-GPU environment problem cannot stop the rest of an
  `-x` session; runs whenever the box shows >= 2 GPUs; N = min(device_count, 8)): one process per GPU, the same cases, every rank
  requires gathered == unsharded bit for bit and an all_reduce proves RCCL saw N ranks.
Each rank is its own process (tests/rccl_worker.py): a RCCL failure cannot take the pytest process down, and a hang is bounded."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_worker.py')


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def launch(world, timeout=600):
    port = _free_port()
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        logs.append((tempfile.TemporaryFile('w+'), tempfile.TemporaryFile('w+')))        # files, not pipes: nobody drains rank k while rank 0 is awaited
        procs.append(subprocess.Popen([sys.executable, WORKER], env=env, stdout=logs[-1][0], stderr=logs[-1][1], text=True))
    reports = []
    try:
        for r, p in enumerate(procs):
            p.wait(timeout=timeout)
            logs[r][0].seek(0); logs[r][1].seek(0)
            out, err = logs[r][0].read(), logs[r][1].read()
            lines = [ln for ln in out.splitlines() if ln.startswith('{"rank"')]
            assert p.returncode == 0 and lines, f'rank {r} exited with {p.returncode}\n--- stdout\n{out[-2000:]}\n--- stderr\n{err[-4000:]}'
            reports.append(json.loads(lines[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return reports


def check(reports, world):
    assert len(reports) == world
    for rep in reports:
        assert rep['ok'] and rep['backend'] == 'nccl' and rep['world'] == world
        assert [c['case'] for c in rep['cases']] == ['C3-weak', 'C3-strong', 'C4-strong', 'C5-strong', 'tiny']
        assert all(c['equal'] for c in rep['cases'])
    # every rank holds the same gathered batch (same reject-code histogram of the same global batch)
    assert all(rep['cases'] == reports[0]['cases'] for rep in reports)


def test_one_rank_rccl_all_gather_equals_unsharded(cuda_device):
    check(launch(1), 1)
