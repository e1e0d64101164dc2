export TMPDIR=/tmp
O=gpurun_out/r4cull; rm -rf $O; mkdir -p $O
timeout 200 python scripts/time_filter.py > $O/filter.txt 2>&1; grep "grid=True" $O/filter.txt
timeout 900 python -m pytest tests/test_collision_gpu.py tests/test_hostprep_gpu.py tests/test_fullsize_properties_gpu.py tests/test_workload_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq_filter.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq_filter.csv > /dev/null; rm -rf $O/pmc_sq
