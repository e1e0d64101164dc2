export TMPDIR=/tmp
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_collision_gpu.py tests/test_fullsize_properties_gpu.py tests/test_workload_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python scripts/time_filter.py > $O/filter.txt 2>&1
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_filter.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_filter.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_filter.py 10 > $O/ktrace.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv "kernel" > /dev/null
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
