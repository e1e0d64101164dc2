import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('this test is marked gpu but no HIP device is visible')
    return torch.device('cuda:0')


# Every GPU test module that runs the PointNet kernels is executed under ALL arithmetic modes of the engine (exact-f32 MFMA, the
# split-half "f16x3" MFMA path, its bf16 sibling, and the 2-unit "f16fp8x2" mode), against the same oracle and bar.
_NET_MODULES = ('test_pointnet_gpu', 'test_predicter_gpu', 'test_pipeline_gpu', 'test_fullsize_properties_gpu', 'test_workload_gpu', 'test_zz_c1_config_gpu',
                'test_pointnet_blocks_gpu')


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split('.')[-1] in _NET_MODULES and 'mlp_precision' in metafunc.fixturenames:
        metafunc.parametrize('mlp_precision', ['f32', 'bf16x3', 'f16x3', 'f16fp8x2'], indirect=True)


@pytest.fixture(autouse=True)
def mlp_precision(request):
    if request.module.__name__.split('.')[-1] not in _NET_MODULES:
        yield None
        return
    from catgrasp_amd import engine
    old = engine.PRECISION
    engine.set_precision(getattr(request, 'param', old))
    yield engine.PRECISION
    engine.set_precision(old)
