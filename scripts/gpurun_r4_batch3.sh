# round 4, third batch: few-row GEMM with its loads in flight, PointNetCls forward as one C call, small-call latency again,
# PMC / trace of the filter kernels as shipped
export TMPDIR=/tmp
O=gpurun_out/r4batch3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_primitives_gpu.py tests/test_pointnet_gpu.py tests/test_pointnet_blocks_gpu.py tests/test_predicter_gpu.py tests/test_collision_gpu.py tests/test_aligning_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1
CATGRASP_AMD_GEMM_SMALL_TILES=100000 timeout 200 python scripts/gemm_small_time.py > $O/gemm_small.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr1 -- python scripts/prof_predict_small2.py 1 device > $O/tr1.log 2>&1
find $O/tr1 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_1.csv \; ; rm -rf $O/tr1
python scripts/time_heads_draw.py > $O/heads_draw.txt 2>&1
timeout 200 python scripts/time_filter.py > $O/filter.txt 2>&1
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_filter.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_filter.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_filter.py 10 > $O/ktrace.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv "filter_grasp\|compose" > /dev/null
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
