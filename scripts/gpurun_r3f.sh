#!/bin/bash
# round 3, call F: set-abstraction register kernel with the asynchronous gather
export TMPDIR=/tmp
O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_primitives_gpu.py -x -q -k "set_abstraction" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
R='sa_group_mlp_max|sa_reg_kernel'
for cfg in "0 0" "0 1" "4 0"; do
  set -- $cfg
  export CATGRASP_SA_WAVES=$1 CATGRASP_SA_DEPHASE=$2
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$R" --output-format csv -d $O/kt -- python scripts/pmc_kernels.py 10 > $O/kt.log 2>&1
  python scripts/pmc_summary.py $O/kt $O/kt_w$1_d$2.csv "kernel" > /dev/null; echo "== waves $1 dephase $2"; grep -E "sa_" $O/kt_w$1_d$2.csv; rm -rf $O/kt
done
unset CATGRASP_SA_WAVES; export CATGRASP_SA_DEPHASE=0
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq -- python scripts/pmc_kernels.py > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq $O/pmc_sq.csv > /dev/null; grep -E "sa_" $O/pmc_sq.csv; rm -rf $O/pmc_sq
