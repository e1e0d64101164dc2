#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python scripts/time_predict_batch.py > $O/run.log 2>&1
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name '*kernel_trace.csv' -exec cp {} $O/ktrace.csv \;
rm -rf $O/stats
head -14 $O/kernel_stats.csv | cut -c1-160
