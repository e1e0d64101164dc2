#!/bin/bash
# The GPU-box jobs of the build rounds, one parameterised script:  gpurun --timeout T -- 'bash scripts/gpu_job.sh <job> [args]'
# Everything is written under gpurun_out/<job>/ ; summaries to be judged are copied to profiles/ afterwards.
export TMPDIR=/tmp
JOB=${1:-suite}; shift
O=gpurun_out/$JOB; rm -rf $O; mkdir -p $O
stats() {   # stats <name> <cmd...>: rocprofv3 kernel statistics of a command -> $O/<name>_kernel_stats.csv
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_st -- "$@" > $O/${name}_trace.log 2>&1
  find $O/_st -name '*kernel_stats.csv' -exec cp {} $O/${name}_kernel_stats.csv \; ; rm -rf $O/_st
}
case $JOB in
  suite)      # the whole GPU suite + smoke
    timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
  tests)      # selected test files / -k expressions:  tests <pytest args>
    timeout 1500 python -m pytest "$@" -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log ;;
  encoder)    # the PointNet++ encoder: tests, stage table, kernel statistics
    timeout 900 python -m pytest tests/test_pointnet2_encoder_gpu.py tests/test_primitives_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
    timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder.json > $O/pp_encoder.txt 2>&1; tail -4 $O/pp_encoder.txt
    timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder_msg.json --msg > $O/pp_encoder_msg.txt 2>&1; tail -3 $O/pp_encoder_msg.txt
    stats pp_encoder python scripts/pp_encoder_profile.py --trace
    stats pp_encoder_msg python scripts/pp_encoder_profile.py --trace --msg
    head -12 $O/pp_encoder_kernel_stats.csv | cut -c1-160 ;;
  bench)      # the default bench line + its kernel statistics
    ( time timeout 900 python bench.py "$@" ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; head -c 600 $O/bench.json
    stats bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection ;;
  pick)       # the pick cycle block of the bench alone (api.pick_cycle) + the tests of the pipeline
    timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_collision_gpu.py tests/test_dataparallel_gpu.py tests/test_pointnet2_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest --no-projection --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench.json'));print(json.dumps(d['api']['pick_cycle'],indent=1))" | head -80 ;;
  small)      # small-call latency with / without the channel split + the tests that pin its bits + the multi-rank dev runs
    timeout 900 python -m pytest tests/test_pointnet_gpu.py tests/test_pointnet_blocks_gpu.py tests/test_predicter_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1; tail -1 $O/predict_small.txt > $O/predict_small.json
    CATGRASP_AMD_POINTMLP_CSPLIT=1 timeout 300 python scripts/time_predict_small.py > $O/predict_small_nosplit.txt 2>&1
    grep candidates $O/predict_small.txt | head -12; echo ---; grep candidates $O/predict_small_nosplit.txt | head -12
    timeout 900 python -m pytest tests/test_bench_multirank_gpu.py -m gpu -x -q > $O/pytest_multirank.log 2>&1; tail -5 $O/pytest_multirank.log ;;
  satile)     # the LDS-tile set-abstraction kernel alone: shipped build, ablation builds (build_abl/lib_sat_*.so), resident-workgroup knob
    timeout 300 python scripts/sa_tile_time.py $O/sa_tile.json > $O/sa_tile.txt 2>&1; grep ssg_sa2 $O/sa_tile.txt
    for lib in build_abl/lib_sat_*.so; do t=$(basename $lib .so); CATGRASP_AMD_LIB=$PWD/$lib timeout 300 python scripts/sa_tile_time.py $O/sa_tile_$t.json > $O/sa_tile_$t.txt 2>&1; echo $t; grep ssg_sa2 $O/sa_tile_$t.txt; done
    CATGRASP_AMD_SAT_PRIO=0 timeout 300 python scripts/sa_tile_time.py $O/sa_tile_noprio.json > $O/sa_tile_noprio.txt 2>&1; echo no prio; grep ssg_sa2 $O/sa_tile_noprio.txt
    for pc in 1 2; do CATGRASP_AMD_SAT_PER_CU=$pc timeout 300 python scripts/sa_tile_time.py $O/sa_tile_percu$pc.json > $O/sa_tile_percu$pc.txt 2>&1; echo per_cu $pc; grep ssg_sa2 $O/sa_tile_percu$pc.txt; done
    timeout 600 python -m pytest tests/test_pointnet2_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
  filter)     # the collision filter: parity tests, timing of the call shapes, issue-side counters
    timeout 900 python -m pytest tests/test_collision_gpu.py tests/test_fullsize_properties_gpu.py tests/test_workload_gpu.py -m gpu -x -q -k "not f16 and not bf16" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    timeout 200 python scripts/time_filter.py > $O/filter_time.txt 2>&1; tail -12 $O/filter_time.txt
    SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
    timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex 'filter_grasp_pose|compose_grasp' --output-format csv -d $O/pmc_sq -- python scripts/pmc_filter.py > $O/pmc_sq_filter.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --kernel-include-regex 'filter_grasp_pose|compose_grasp' --output-format csv -d $O/ktrace -- python scripts/pmc_filter.py 10 > $O/ktrace_filter.log 2>&1
    python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq_filter.csv > /dev/null; python scripts/pmc_summary.py $O/ktrace $O/ktrace_filter.csv "grasp_pose" > /dev/null
    rm -rf $O/pmc_sq $O/ktrace; grep "true" $O/pmc_sq_filter.csv; cat $O/ktrace_filter.csv ;;
  final)      # the measurements of the shipped state: GPU suite, smoke, bench line (+ kernel statistics), helper table, encoder, C4 / C5
    timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
    stats bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection
    timeout 300 python scripts/hbm_kernels.py > $O/hbm_kernels.json 2> $O/hbm_kernels.err; tail -2 $O/hbm_kernels.err
    timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder.json > $O/pp_encoder.txt 2>&1
    timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder_msg.json --msg > $O/pp_encoder_msg.txt 2>&1
    stats pp_encoder python scripts/pp_encoder_profile.py --trace
    stats pp_encoder_msg python scripts/pp_encoder_profile.py --trace --msg
    timeout 300 python scripts/sa_tile_time.py $O/sa_tile.json > $O/sa_tile.txt 2>&1
    ( time timeout 600 python bench.py --gpus 1 --workload C4 --steps 3 --warmup 1 --secondary "" --no-api --no-cpu-baseline ) > $O/bench_c4_n1.json 2> $O/bench_c4_n1.err
    ( time timeout 600 python bench.py --gpus 1 --workload C5 --steps 3 --warmup 1 --secondary "f32" --no-api --no-cpu-baseline ) > $O/bench_c5_n1.json 2> $O/bench_c5_n1.err
    ls -la $O; head -c 600 $O/bench.json ;;
  py)         # any script:  py scripts/x.py args...
    timeout 1200 python "$@" > $O/out.txt 2>&1; tail -40 $O/out.txt ;;
  *) echo "unknown job $JOB"; exit 2 ;;
esac
