#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/shuffle; rm -rf $O; mkdir -p $O
timeout 100 python -m pytest tests/test_hostprep_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 90 python scripts/time_predict_batch.py > $O/predict_batch.json 2> $O/predict_batch.err; cat $O/predict_batch.json; tail -2 $O/predict_batch.err
