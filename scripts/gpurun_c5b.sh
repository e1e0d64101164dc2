#!/bin/bash
# C5: rocprofv3 kernel statistics of the bench command on one GPU, and the 2-rank dev run (two ranks on ONE device, gloo) of the same workload
export TMPDIR=/tmp
O=gpurun_out/c5b; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --workload C5 --steps 2 --warmup 1 --secondary "" --no-api --no-cpu-baseline > $O/bench_c5_profiled.json 2> $O/bench_c5_profiled.err
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
head -8 $O/kernel_stats.csv
CATGRASP_BENCH_BACKEND=gloo CATGRASP_BENCH_DEVICE=0 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload C5 --candidates-total 100000 --steps 2 --warmup 1 --secondary "" > $O/bench_c5_2rank.json 2> $O/bench_c5_2rank.err
tail -c 400 $O/bench_c5_2rank.json; tail -2 $O/bench_c5_2rank.err
