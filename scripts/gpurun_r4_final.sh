#!/bin/bash
# round 4: the measurements of the shipped state -- GPU suite, smoke, bench line (+ its rocprofv3 kernel statistics), helper-kernel table,
# PMC passes (filter kernels; set abstraction / FPS), API timing, small-call latency, SA layer table, hypothesis draw, C4 / C5 lines
export TMPDIR=/tmp
O=gpurun_out/r4final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection > $O/bench_profiled.json 2> $O/bench_profiled.err
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
timeout 300 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm_kernels.err; tail -2 $O/hbm_kernels.err
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq_filter.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_fetch -- python scripts/pmc_filter.py > $O/pmc_fetch_filter.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$R" --output-format csv -d $O/pmc_write -- python scripts/pmc_filter.py > $O/pmc_write_filter.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/ktrace -- python scripts/pmc_filter.py 10 > $O/ktrace_filter.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq_filter.csv > /dev/null; python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch_filter.csv > /dev/null
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write_filter.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace_filter.csv "grasp_pose" > /dev/null
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ktrace
timeout 200 python scripts/time_filter.py > $O/filter.txt 2>&1
timeout 300 python scripts/time_predict_batch.py > $O/predict_batch_api.json 2> /dev/null
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1
timeout 300 python scripts/sa_layer_time.py $O/sa_layer.json > $O/sa_layer.txt 2>&1
python scripts/time_heads_draw.py > $O/heads_draw.txt 2>&1
timeout 100 python scripts/gemm_small_time.py > $O/gemm_small_default.txt 2>&1
CATGRASP_AMD_GEMM_SMALL_TILES=0 timeout 100 python scripts/gemm_small_time.py > $O/gemm_small_tile_kernel.txt 2>&1
( time timeout 400 python bench.py --gpus 1 --workload C4 --steps 3 --warmup 1 --secondary "" --no-api --no-cpu-baseline ) > $O/bench_c4_n1.json 2> $O/bench_c4_n1.err
( time timeout 400 python bench.py --gpus 1 --workload C5 --steps 3 --warmup 1 --secondary "f32" --no-api --no-cpu-baseline ) > $O/bench_c5_n1.json 2> $O/bench_c5_n1.err
ls -la $O; head -c 400 $O/bench.json
