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
    stats bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs ;;
  pick)       # the pick cycle block of the bench alone (api.pick_cycle) + the tests of the pipeline
    timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_collision_gpu.py tests/test_dataparallel_gpu.py tests/test_pointnet2_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench.json'));print(json.dumps(d['api']['pick_cycle'],indent=1))" | head -80 ;;
  small)      # small-call latency with / without the channel split + the tests that pin its bits + the multi-rank dev runs
    timeout 900 python -m pytest tests/test_pointnet_gpu.py tests/test_pointnet_blocks_gpu.py tests/test_predicter_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1; tail -1 $O/predict_small.txt > $O/predict_small.json
    CATGRASP_AMD_POINTMLP_CSPLIT=1 timeout 300 python scripts/time_predict_small.py > $O/predict_small_nosplit.txt 2>&1
    grep candidates $O/predict_small.txt | head -12; echo ---; grep candidates $O/predict_small_nosplit.txt | head -12
    timeout 900 python -m pytest tests/test_bench_multirank_gpu.py -m gpu -x -q > $O/pytest_multirank.log 2>&1; tail -5 $O/pytest_multirank.log ;;
  satile)     # the LDS-tile set-abstraction kernel alone: shipped build, tile-rows / resident-workgroup knobs, ablation builds (build_abl/lib_sat_*.so)
    timeout 300 python scripts/sa_tile_time.py $O/sa_tile.json > $O/sa_tile.txt 2>&1; grep ssg_sa2 $O/sa_tile.txt
    for tr in 64 128; do CATGRASP_AMD_SAT_TILE_ROWS=$tr timeout 300 python scripts/sa_tile_time.py $O/sa_tile_rows$tr.json > $O/sa_tile_rows$tr.txt 2>&1; echo tile rows $tr; grep ssg_sa2 $O/sa_tile_rows$tr.txt; done
    for lib in build_abl/lib_sat_*.so; do [ -f $lib ] || continue; t=$(basename $lib .so); CATGRASP_AMD_LIB=$PWD/$lib timeout 300 python scripts/sa_tile_time.py $O/sa_tile_$t.json > $O/sa_tile_$t.txt 2>&1; echo $t; grep ssg_sa2 $O/sa_tile_$t.txt; done
    CATGRASP_AMD_SAT_PER_CU=1 timeout 300 python scripts/sa_tile_time.py $O/sa_tile_percu1.json > $O/sa_tile_percu1.txt 2>&1; echo per_cu 1; grep ssg_sa2 $O/sa_tile_percu1.txt
    timeout 600 python -m pytest tests/test_pointnet2_encoder_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
  pmcsat)     # issue-side counters of the tile set-abstraction kernel (64 clouds), 64- and 128-row tiles
    for tr in 64 128; do
      C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
      C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
      CATGRASP_AMD_SAT_TILE_ROWS=$tr timeout 300 rocprofv3 --pmc $C1 --kernel-include-regex sa_tile --output-format csv -d $O/p1_$tr -- python scripts/pmc_sa_tile.py > $O/p1_$tr.log 2>&1
      CATGRASP_AMD_SAT_TILE_ROWS=$tr timeout 300 rocprofv3 --pmc $C2 --kernel-include-regex sa_tile --output-format csv -d $O/p2_$tr -- python scripts/pmc_sa_tile.py > $O/p2_$tr.log 2>&1
      CATGRASP_AMD_SAT_TILE_ROWS=$tr timeout 300 rocprofv3 --kernel-trace --kernel-include-regex sa_tile --output-format csv -d $O/kt_$tr -- python scripts/pmc_sa_tile.py > $O/kt_$tr.log 2>&1
      python scripts/pmc_summary.py $O/p1_$tr $O/pmc1_rows$tr.csv > /dev/null; python scripts/pmc_summary.py $O/p2_$tr $O/pmc2_rows$tr.csv > /dev/null; python scripts/pmc_summary.py $O/kt_$tr $O/kt_rows$tr.csv > /dev/null
      rm -rf $O/p1_$tr $O/p2_$tr $O/kt_$tr; cat $O/pmc1_rows$tr.csv $O/pmc2_rows$tr.csv $O/kt_rows$tr.csv; tail -2 $O/p2_$tr.log
    done ;;
  check)      # the driver's round-end sequence on the final tree: GPU suite, smoke, the default bench line
    timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 300 $O/bench.err; head -c 500 $O/bench_default_flags.json ;;
  r6a)        # round 6, first contact: stricter encoder tests, refactored bench (multirank + the default line with `configs`)
    timeout 1200 python -m pytest tests/test_pointnet2_encoder_gpu.py tests/test_primitives_gpu.py tests/test_bench_multirank_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
    ( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 600 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench_default_flags.json'));print(json.dumps(d['timing_s'],indent=1));print(json.dumps({k:{kk:v.get(kk) for kk in ("value","ms_per_step","wall_s","error")} for k,v in d['configs'].items()}));print(d['value'],d['ms_per_step'])" ;;
  r6b)        # round 6: the one-launch filter (FilterPlan) -- parity tests, the step, the default line
    timeout 1500 python -m pytest tests/test_collision_gpu.py tests/test_workload_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_bench_multirank_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log
    ( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 400 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench_default_flags.json'));print(json.dumps(d['timing_s'],indent=1));print(json.dumps({k:{kk:v.get(kk) for kk in ('value','ms_per_step','wall_s','error')} for k,v in d['configs'].items()}));print(d['value'],d['ms_per_step']);print(json.dumps({k:v for k,v in d['roofline_filter'].items() if k not in ('note','cache_level')}));print(json.dumps(d['api']['pick_cycle']['default']))" ;;
  pmcsplit)   # issue-side counters of the split-precision encoder-pass kernels (C5's dominant kernel: <2, 8, false, false> = bf16x3), 4,096 candidates
    C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
    C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
    C3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
    for prec in bf16x3 f32; do
      i=0; for C in "$C1" "$C2" "$C3"; do i=$((i+1))
        timeout 300 rocprofv3 --pmc $C --kernel-include-regex pointmlp_max --output-format csv -d $O/p${i}_$prec -- python scripts/quick_cls.py 4096 $prec > $O/p${i}_$prec.log 2>&1
        python scripts/pmc_summary.py $O/p${i}_$prec $O/pmc${i}_$prec.csv > /dev/null; rm -rf $O/p${i}_$prec
      done
      timeout 300 rocprofv3 --kernel-trace --kernel-include-regex pointmlp_max --output-format csv -d $O/kt_$prec -- python scripts/quick_cls.py 4096 $prec > $O/kt_$prec.log 2>&1
      python scripts/pmc_summary.py $O/kt_$prec $O/kt_$prec.csv > /dev/null; rm -rf $O/kt_$prec; tail -1 $O/kt_$prec.log
      cat $O/pmc1_$prec.csv $O/pmc2_$prec.csv $O/pmc3_$prec.csv $O/kt_$prec.csv | grep -v "^kernel" | grep "kernel<2" > $O/pmc_sq_pointmlp_$prec.csv; cat $O/pmc_sq_pointmlp_$prec.csv
    done ;;
  filterplan) # the step's filter as one launch sequence: in-tree build vs ablation builds (build_abl/lib_*.so), + issue counters of each
    for lib in in-tree build_abl/lib_*.so; do
      [ $lib = in-tree ] && unset CATGRASP_AMD_LIB || export CATGRASP_AMD_LIB=$PWD/$lib
      t=$(basename $lib .so)
      timeout 300 python scripts/time_filter_plan.py > $O/time_$t.txt 2>&1; tail -1 $O/time_$t.txt
      timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-include-regex filter_grasp_pose_kernel --output-format csv -d $O/p_$t -- python scripts/time_filter_plan.py --pmc > $O/p_$t.log 2>&1
      python scripts/pmc_summary.py $O/p_$t $O/pmc_$t.csv > /dev/null; rm -rf $O/p_$t; grep "true, true" $O/pmc_$t.csv
    done; unset CATGRASP_AMD_LIB
    timeout 900 python -m pytest tests/test_collision_gpu.py tests/test_workload_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
  r6c)        # the chain kernel + census: tests, small-call latency, encoder stage table
    timeout 1500 python -m pytest tests/test_pointnet_gpu.py tests/test_pointnet2_encoder_gpu.py tests/test_predicter_gpu.py tests/test_bench_multirank_gpu.py tests/test_zzz_rccl_multirank_gpu.py tests/test_pointnet_blocks_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -8 $O/pytest.log
    timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1; tail -1 $O/predict_small.txt > $O/predict_small.json; grep candidates $O/predict_small.txt | head -12
    CATGRASP_AMD_GEMM_CHAIN=0 timeout 300 python scripts/time_predict_small.py > $O/predict_small_nochain.txt 2>&1; echo --- no chain; grep candidates $O/predict_small_nochain.txt | head -12
    timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder.json > $O/pp_encoder.txt 2>&1; tail -4 $O/pp_encoder.txt
    CATGRASP_AMD_GEMM_CHAIN=0 timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder_nochain.json > $O/pp_encoder_nochain.txt 2>&1; tail -4 $O/pp_encoder_nochain.txt
    stats pp_encoder python scripts/pp_encoder_profile.py --trace
    head -14 $O/pp_encoder_kernel_stats.csv | cut -c1-170 ;;
  r6d)        # tickets in the filter, cheaper chain barrier, overlapped pick cycle
    timeout 300 python scripts/time_filter_plan.py > $O/time_tickets.txt 2>&1; tail -1 $O/time_tickets.txt
    CATGRASP_AMD_FILTER_STATIC=1 timeout 300 python scripts/time_filter_plan.py > $O/time_static.txt 2>&1; tail -1 $O/time_static.txt
    timeout 300 python scripts/time_filter_plan.py 400000 > $O/time_tickets_400k.txt 2>&1; tail -1 $O/time_tickets_400k.txt
    CATGRASP_AMD_FILTER_STATIC=1 timeout 300 python scripts/time_filter_plan.py 400000 > $O/time_static_400k.txt 2>&1; tail -1 $O/time_static_400k.txt
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-include-regex filter_grasp_pose_kernel --output-format csv -d $O/p_t -- python scripts/time_filter_plan.py --pmc > $O/p_t.log 2>&1
    python scripts/pmc_summary.py $O/p_t $O/pmc_tickets.csv > /dev/null; rm -rf $O/p_t; grep "true, true" $O/pmc_tickets.csv
    timeout 1500 python -m pytest tests/test_collision_gpu.py tests/test_workload_gpu.py tests/test_pipeline_gpu.py tests/test_pointnet_gpu.py tests/test_pointnet2_encoder_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -8 $O/pytest.log
    CATGRASP_AMD_GEMM_CHAIN=1 timeout 300 python scripts/time_predict_small.py > $O/predict_small_chain.txt 2>&1; echo --- chain; grep candidates $O/predict_small_chain.txt | head -4
    CATGRASP_AMD_GEMM_CHAIN=0 timeout 300 python scripts/time_predict_small.py > $O/predict_small_nochain.txt 2>&1; echo --- no chain; grep candidates $O/predict_small_nochain.txt | head -4
    CATGRASP_AMD_GEMM_CHAIN=1 timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder_chain.json > $O/pp_encoder_chain.txt 2>&1; grep -o '"clouds": [0-9]*\|"encoder_ms": [0-9.]*\|gemm_chain": [0-9.]*' $O/pp_encoder_chain.txt | head -12
    CATGRASP_AMD_GEMM_CHAIN=0 timeout 300 python scripts/pp_encoder_profile.py $O/pp_encoder_nochain.json > $O/pp_encoder_nochain.txt 2>&1; grep -o '"clouds": [0-9]*\|"encoder_ms": [0-9.]*\|gemm_chain": [0-9.]*' $O/pp_encoder_nochain.txt | head -12
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench.json'));p=d['api']['pick_cycle'];print(json.dumps({k:(v['wall_ms_per_object'] if isinstance(v,dict) and 'wall_ms_per_object' in v else v) for k,v in p.items() if k!='note'},indent=1)[:1500]);print(json.dumps(p['default']['ms_per_object_by_stage']))" ;;
  r6e)        # filter workgroups-per-CU knob, overlapped pick cycle with wait instrumentation
    for cap in 16 32 64 4096; do echo cap $cap; CATGRASP_AMD_FILTER_BLOCKS_PER_CU=$cap timeout 300 python scripts/time_filter_plan.py > $O/time_cap$cap.txt 2>&1; tail -1 $O/time_cap$cap.txt | cut -c1-200; done
    CATGRASP_AMD_FILTER_BLOCKS_PER_CU=4096 timeout 300 python scripts/time_filter_plan.py 400000 > $O/time_cap4096_400k.txt 2>&1; tail -1 $O/time_cap4096_400k.txt | cut -c1-200
    timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_collision_gpu.py tests/test_pointnet_gpu.py tests/test_pointnet2_encoder_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench.json'));p=d['api']['pick_cycle'];print(json.dumps({k:(v['wall_ms_per_object'] if isinstance(v,dict) and 'wall_ms_per_object' in v else v) for k,v in p.items() if k!='note'},indent=1)[:1500]);print(json.dumps(p['default']['ms_per_object_by_stage']))" ;;
  r6f)        # split kernel without SLP-packed f32 ops; stages-thread instrumentation; filter cap at 400k
    for i in 1 2 3; do
      timeout 120 python scripts/quick_cls.py 4096 bf16x3 2>/dev/null | tail -1
      CATGRASP_AMD_LIB=$PWD/build_abl/lib_split_noslp.so timeout 120 python scripts/quick_cls.py 4096 bf16x3 2>/dev/null | tail -1 | sed 's/^/noslp: /'
    done | tee $O/split_noslp.txt
    timeout 300 python scripts/time_filter_plan.py 400000 > $O/time_cap32_400k.txt 2>&1; tail -1 $O/time_cap32_400k.txt | cut -c1-200
    CATGRASP_AMD_FILTER_BLOCKS_PER_CU=16 timeout 300 python scripts/time_filter_plan.py 400000 > $O/time_cap16_400k.txt 2>&1; tail -1 $O/time_cap16_400k.txt | cut -c1-200
    timeout 300 python scripts/time_filter_plan.py > $O/time_cap32.txt 2>&1; tail -1 $O/time_cap32.txt | cut -c1-200
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs --secondary "" > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
    python -c "import json;d=json.load(open('$O/bench.json'));p=d['api']['pick_cycle'];print(json.dumps({k:(v['wall_ms_per_object'] if isinstance(v,dict) and 'wall_ms_per_object' in v else v) for k,v in p.items() if k!='note'},indent=1)[:1500]);print(json.dumps(p['default']['ms_per_object_by_stage']))" ;;
  r6lines)    # the round's bench lines: default flags (C3 + configs), C4 and C5 alone, kernel statistics of the default step
    ( time timeout 900 python bench.py ) > $O/bench_line.json 2> $O/bench.err; tail -c 200 $O/bench.err
    timeout 900 python bench.py --workload C4 --no-api --no-cpu-baseline --no-rccl-selftest --secondary "" > $O/bench_line_c4_n1.json 2> $O/c4.err; head -c 300 $O/bench_line_c4_n1.json; echo
    timeout 900 python bench.py --workload C5 --no-api --no-cpu-baseline --no-rccl-selftest --secondary "" > $O/bench_line_c5_n1.json 2> $O/c5.err; head -c 300 $O/bench_line_c5_n1.json; echo
    stats bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection --no-configs
    stats bench_c5 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest --no-projection --secondary ""
    python -c "import json;d=json.load(open('$O/bench_line.json'));print(json.dumps(d['timing_s']));print(d['value'],d['ms_per_step'],{k:(v.get('value'),v.get('ms_per_step'),v.get('wall_s')) for k,v in d['configs'].items()});print(d['api']['predict_batch_small_calls']);print(d['api']['pick_cycle']['default']['wall_ms_per_object'])" ;;
  r6final)    # the driver's round-end sequence on the final tree (suite, smoke, default bench) + the C4 / C5 lines and the example in the same (warm) state
    timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 200 $O/bench.err
    timeout 900 python bench.py --workload C4 --no-api --no-cpu-baseline --no-rccl-selftest --secondary "" > $O/bench_line_c4_n1.json 2> $O/c4.err
    timeout 900 python bench.py --workload C5 --no-api --no-cpu-baseline --no-rccl-selftest --secondary "" > $O/bench_line_c5_n1.json 2> $O/c5.err
    timeout 300 python examples/run_scene.py > $O/example.txt 2>&1; tail -3 $O/example.txt
    python -c "import json;d=json.load(open('$O/bench_default_flags.json'));c4=json.load(open('$O/bench_line_c4_n1.json'));c5=json.load(open('$O/bench_line_c5_n1.json'));print(d['value'],d['ms_per_step'],{k:v['value'] for k,v in d['configs'].items()},'standalone',c4['value'],c5['value'],d['api']['pick_cycle']['default']['wall_ms_per_object'],d['api']['predict_batch_small_calls'][:1],d['roofline_filter']['frac'],d['timing_s']['total'])" ;;
  py)         # any script:  py scripts/x.py args...
    timeout 1200 python "$@" > $O/out.txt 2>&1; tail -40 $O/out.txt ;;
  *) echo "unknown job $JOB"; exit 2 ;;
esac
