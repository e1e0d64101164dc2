#!/bin/bash
# round 3, call B: the new GPU tests (holes closed, kd-tree RANSAC branch, pointgroup fixes) + PMC baseline of the north-star kernels per grid size
export TMPDIR=/tmp
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_pointnet_blocks_gpu.py tests/test_aligning_gpu.py tests/test_pointgroup_ops_gpu.py tests/test_primitives_gpu.py "tests/test_predicter_gpu.py::test_standalone_building_blocks_run_on_the_hip_passes" "tests/test_predicter_gpu.py::test_nunocs_predict_nocs" -x -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='sa_group_mlp_max|fps_kernel|filter_grasp_pose'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_kernels.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_kernels.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_kernels.py > $O/ktrace.log 2>&1
head -3 $(find $O/pmc_sq -name '*counter_collection.csv' | head -1)
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv "kernel" > /dev/null
cat $O/pmc_sq.csv $O/pmc_fetch.csv $O/pmc_write.csv; grep -E "sa_group|fps_kernel|filter_grasp" $O/ktrace.csv
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
