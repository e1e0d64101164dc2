#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/bench_final; rm -rf $O; mkdir -p $O
timeout 230 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -2 $O/bench.err
