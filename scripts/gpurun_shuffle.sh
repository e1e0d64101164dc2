#!/bin/bash
# the device swap chain of the numpy-exact resampling draw: parity tests, then the api block of the bench
export TMPDIR=/tmp
O=gpurun_out/shuffle; rm -rf $O; mkdir -p $O
timeout 240 python -m pytest tests/test_hostprep_gpu.py tests/test_predicter_gpu.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
