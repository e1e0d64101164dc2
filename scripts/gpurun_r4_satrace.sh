#!/bin/bash
# round 4: per-launch kernel durations of the set-abstraction layer (one cloud / 16 clouds), rocprofv3 kernel trace
export TMPDIR=/tmp
O=gpurun_out/r4satrace; rm -rf $O; mkdir -p $O
timeout 80 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python scripts/sa_layer_trace.py > $O/trace.log 2>&1
python scripts/pmc_summary.py $O/kt $O/sa_layer_kernel_durations.csv "anonymous" > /dev/null; rm -rf $O/kt; cat $O/sa_layer_kernel_durations.csv
