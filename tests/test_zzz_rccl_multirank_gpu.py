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
