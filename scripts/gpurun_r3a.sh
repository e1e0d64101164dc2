#!/bin/bash
# round 3, call A: RCCL self-test, bench with its default counter passes + one-rank RCCL block, PMC baseline of the north-star kernels
export TMPDIR=/tmp
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed_rccl_gpu.py -x -q > $O/pytest_rccl.log 2>&1; tail -15 $O/pytest_rccl.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-api --no-cpu-baseline --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='sa_group_mlp_max|farthest_point_sample|filter_grasp_pose'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_kernels.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_kernels.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_kernels.py > $O/ktrace.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv > /dev/null
cat $O/pmc_sq.csv $O/pmc_fetch.csv $O/pmc_write.csv $O/ktrace.csv
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
ls -la $O
