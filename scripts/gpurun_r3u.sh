#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --candidates 6250 --steps 3 --warmup 2 --secondary "" --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest > $O/b.json 2> $O/b.err
find $O/tr -name '*kernel_trace.csv' -exec cp {} $O/ktrace.csv \;
rm -rf $O/tr; ls -la $O
