#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hostprep_gpu.py "tests/test_predicter_gpu.py::test_nunocs_predict_nocs" -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm.err; python -c "
import json; h=json.load(open('$O/hbm_kernels.json'))
for r in h['kernels']:
    if 'nunocs' in r['kernel'] or 'sa_' in r['kernel'] or 'farthest' in r['kernel']: print(r['kernel'][:60], r['avg_launch_us'], r['achieved_GBps'], r['frac_of_8TBps'], r.get('frac_of_157.3_TFLOPs'))"
