#!/bin/bash
# round 4: the swap-partner extraction of the numpy-stream resampling draw on the vectorised walk -- its tests and the API rate
export TMPDIR=/tmp
O=gpurun_out/r4partners; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hostprep_gpu.py tests/test_predicter_gpu.py tests/test_zz_c1_config_gpu.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python scripts/time_predict_batch.py > $O/predict_batch_api.json 2> /dev/null; cat $O/predict_batch_api.json | head -c 900
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1; grep candidates $O/predict_small.txt | head -8
