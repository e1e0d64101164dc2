#!/bin/bash
# round 3, call G: FPS with unsigned-bit compares: parity + duration
export TMPDIR=/tmp
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_primitives_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 0 1; do
CATGRASP_FPS_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "fps_kernel" --output-format csv -d $O/kt -- python scripts/pmc_kernels.py 4 > $O/kt.log 2>&1
python scripts/pmc_summary.py $O/kt $O/kt_fps_$v.csv "kernel" > /dev/null; grep -E "fps_kernel" $O/kt_fps_$v.csv; rm -rf $O/kt
done
