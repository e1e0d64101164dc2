#!/bin/bash
# round 3, call E: set-abstraction register kernel: waves per workgroup x start-up de-phasing, durations by kernel trace
export TMPDIR=/tmp
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_primitives_gpu.py -x -q -k "set_abstraction" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
R='sa_group_mlp_max|sa_reg_kernel'
for cfg in "0 0" "8 0" "8 1" "8 2" "8 3" "12 0" "12 1" "12 2" "11 1" "10 1" "4 1"; do
  set -- $cfg
  export CATGRASP_SA_WAVES=$1 CATGRASP_SA_DEPHASE=$2
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/kt -- python scripts/pmc_kernels.py 10 > $O/kt.log 2>&1
  python scripts/pmc_summary.py $O/kt $O/kt_w$1_d$2.csv "kernel" > /dev/null; echo "== waves $1 dephase $2"; grep -E "sa_" $O/kt_w$1_d$2.csv; rm -rf $O/kt
done
