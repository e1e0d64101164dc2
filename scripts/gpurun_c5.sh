#!/bin/bash
# one short GPU call: the mixed-bin (C5) scene batch tests and the C5 bench line on one GPU
export TMPDIR=/tmp
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 200 python -m pytest tests/test_workload_gpu.py tests/test_distributed_gloo.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 240 python bench.py --gpus 1 --workload C5 --steps 3 --warmup 1 --secondary f32 --no-api --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5.json; tail -3 $O/bench_c5.err
