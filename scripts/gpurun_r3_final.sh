#!/bin/bash
# round 3: the measurements of the shipped state -- GPU suite, smoke, bench line (+ its rocprofv3 kernel statistics), helper-kernel table, PMC passes, API timing
export TMPDIR=/tmp
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest > $O/bench_profiled.json 2> $O/bench_profiled.err
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
timeout 300 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm_kernels.err; tail -2 $O/hbm_kernels.err
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='sa_group_mlp_max|sa_reg_kernel|fps_kernel|filter_grasp_pose'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_kernels.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_kernels.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_kernels.py 10 > $O/ktrace.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace.csv "kernel" > /dev/null
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
timeout 300 python scripts/time_predict_batch.py > $O/predict_batch_api.json 2> /dev/null
( time timeout 400 python bench.py --gpus 1 --workload C4 --steps 3 --warmup 1 --secondary "" --no-api --no-cpu-baseline ) > $O/bench_c4_n1.json 2> $O/bench_c4_n1.err
( time timeout 400 python bench.py --gpus 1 --workload C5 --steps 3 --warmup 1 --secondary "f32" --no-api --no-cpu-baseline ) > $O/bench_c5_n1.json 2> $O/bench_c5_n1.err
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_sa -- python scripts/sa_scaling.py > $O/sa_scaling.log 2>&1
find $O/tr_sa -name '*kernel_trace.csv' -exec cp {} $O/ktrace_sa_scaling.csv \; ; rm -rf $O/tr_sa
timeout 100 python scripts/fps_time.py > $O/fps_time.txt 2> /dev/null
timeout 100 python scripts/sa_time.py > $O/sa_time.txt 2> /dev/null
for n in 50000 25000 12500 6250; do
timeout 300 python bench.py --candidates $n --steps 10 --warmup 3 --secondary "" --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest > $O/shard_$n.json 2> /dev/null
done
ls -la $O; head -c 600 $O/bench.json
