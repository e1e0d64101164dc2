#!/bin/bash
# round 3, call H: predict_batch chunk pipeline: parity tests + wall-clock through the reference entry point
export TMPDIR=/tmp
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_predicter_gpu.py tests/test_zz_c1_config_gpu.py tests/test_hostprep_gpu.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 600 python scripts/time_predict_batch.py > $O/predict_batch_api.json 2> $O/predict_batch_api.err; cat $O/predict_batch_api.json; tail -3 $O/predict_batch_api.err
