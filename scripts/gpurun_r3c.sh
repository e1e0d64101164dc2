#!/bin/bash
# round 3, call C: new set-abstraction kernel (persistent, LDS-resident weights) + FPS v3: parity tests, kernel durations by rocprofv3 kernel trace
export TMPDIR=/tmp
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_primitives_gpu.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
R='sa_group_mlp_max|fps_kernel|filter_grasp_pose'
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_kernels.py 10 > $O/ktrace.log 2>&1
python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv "kernel" > /dev/null; grep -E "sa_group|fps_kernel|filter_grasp" $O/ktrace.csv
CATGRASP_FPS_VARIANT=1 timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "fps_kernel" --output-format csv -d $O/ktrace1 -- python scripts/pmc_kernels.py 4 > $O/ktrace1.log 2>&1
python scripts/pmc_summary.py $O/ktrace1 $O/ktrace_fps_variant1.csv "kernel" > /dev/null; grep -E "fps_kernel" $O/ktrace_fps_variant1.csv
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; grep -E "sa_group|fps_kernel" $O/pmc_sq.csv
rm -rf $O/ktrace $O/ktrace1 $O/pmc_sq
