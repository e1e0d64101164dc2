#!/bin/bash
# round 4: the last host-side changes (one upload per cloud, pointer alignment check of the filter entry point) on the tests that see them
export TMPDIR=/tmp
O=gpurun_out/r4last; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_predicter_gpu.py tests/test_collision_gpu.py tests/test_pipeline_gpu.py tests/test_workload_gpu.py tests/test_zz_c1_config_gpu.py tests/test_aligning_gpu.py tests/test_hostprep_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1; grep candidates $O/predict_small.txt | head -6
