#!/bin/bash
# round 4, after the voxel-block culling of the filter kernel and the register-resident sort of the resampling draw: the tests that
# touch them, their timings / counters, the helper-kernel table, and the bench line again
export TMPDIR=/tmp
O=gpurun_out/r4final2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_collision_gpu.py tests/test_hostprep_gpu.py tests/test_fullsize_properties_gpu.py tests/test_workload_gpu.py tests/test_pipeline_gpu.py tests/test_predicter_gpu.py tests/test_pointnet_gpu.py tests/test_distributed_rccl_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 200 python scripts/time_filter.py > $O/filter.txt 2>&1; grep "grid=True" $O/filter.txt
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq_filter.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_filter.py > $O/pmc_fetch_filter.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_filter.py > $O/pmc_write_filter.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_filter.py 10 > $O/ktrace_filter.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq_filter.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch_filter.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write_filter.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace_filter.csv "grasp_pose" > /dev/null
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
timeout 300 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm_kernels.err; tail -2 $O/hbm_kernels.err
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection > $O/bench_profiled.json 2> $O/bench_profiled.err
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
ls $O; head -c 300 $O/bench.json
