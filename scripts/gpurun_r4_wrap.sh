#!/bin/bash
# round 4, after the farthest-point-sampling / ball-query / set-abstraction-layer work: the GPU tests that reach those kernels, smoke, the
# helper-kernel table, a kernel trace of the layer, and the default bench line on the final tree
export TMPDIR=/tmp
O=gpurun_out/r4wrap; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_primitives_gpu.py tests/test_pointnet_blocks_gpu.py tests/test_pointgroup_ops_gpu.py tests/test_aligning_gpu.py tests/test_zz_c1_config_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 200 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm_kernels.err; tail -2 $O/hbm_kernels.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python scripts/sa_layer_trace.py > $O/sa_trace.log 2>&1
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/sa_layer_kernel_stats.csv \; ; rm -rf $O/stats; head -8 $O/sa_layer_kernel_stats.csv | cut -c1-200
( time timeout 600 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 200 $O/bench.err; head -c 400 $O/bench_default_flags.json
