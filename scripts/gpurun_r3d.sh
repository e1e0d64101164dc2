#!/bin/bash
# round 3, call D: register-resident set-abstraction kernel: parity over the signatures + durations (default lib vs alt lib = 2 waves/SIMD for 64-64-128, vs strip kernel)
export TMPDIR=/tmp
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_primitives_gpu.py -x -q -k "set_abstraction" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
R='sa_group_mlp_max|sa_reg_kernel'
for v in default alt strip; do
  export CATGRASP_AMD_LIB=; unset CATGRASP_AMD_LIB; unset CATGRASP_SA_STRIP
  [ $v = alt ] && export CATGRASP_AMD_LIB=$PWD/catgrasp_amd/libcatgrasp_amd_alt.so
  [ $v = strip ] && export CATGRASP_SA_STRIP=1
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/kt_$v -- python scripts/pmc_kernels.py 10 > $O/kt_$v.log 2>&1
  python scripts/pmc_summary.py $O/kt_$v $O/kt_$v.csv "kernel" > /dev/null; echo "== $v"; grep -E "sa_" $O/kt_$v.csv; rm -rf $O/kt_$v
done
unset CATGRASP_AMD_LIB; unset CATGRASP_SA_STRIP
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; grep -E "sa_" $O/pmc_sq.csv; rm -rf $O/pmc_sq
