#!/bin/bash
# one GPU call: the full GPU suite, smoke, the profiled bench command and the bench line (final state of the round)
export TMPDIR=/tmp
O=gpurun_out/final2; rm -rf $O; mkdir -p $O
timeout 800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api > $O/bench_profiled.json 2> $O/bench_profiled.err
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
ls $O
