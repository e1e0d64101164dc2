#!/bin/bash
# predicter tests after the predict_batch clean-up + a per-rank-sized step (6,250 candidates: what one of 8 ranks runs under strong scaling)
export TMPDIR=/tmp
O=gpurun_out/r3l; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_predicter_gpu.py tests/test_zz_c1_config_gpu.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --candidates 6250 --steps 20 --warmup 5 --secondary "" --no-api --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest > $O/bench_6250.json 2> $O/bench_6250.err
python -c "
import json; d=json.load(open('$O/bench_6250.json')); print('6250 candidates over 8 objects:', d['value'], 'cand/s', d['ms_per_step'], 'ms/step; ideal', round(6250/82400*1e3,2), 'ms')"
