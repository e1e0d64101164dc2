export TMPDIR=/tmp
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
R='filter_grasp_pose|compose_grasp'
for v in $VARIANTS; do
export CATGRASP_AMD_LIB=build_abl/lib_$v.so
for cell in 0.002 0.001; do
export CATGRASP_AMD_GRID_CELL=$cell
echo "== $v cell $cell" >> $O/filter.txt
timeout 300 python scripts/time_filter.py 2>&1 | grep "grid=True" >> $O/filter.txt
done
unset CATGRASP_AMD_GRID_CELL
timeout 300 rocprofv3 --pmc $SQ --kernel-include-regex "$R" --output-format csv -d $O/pmc_sq_$v -- python scripts/pmc_filter.py > $O/pmc_sq_$v.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq_$v $O/pmc_sq_$v.csv > /dev/null
rm -rf $O/pmc_sq_$v
done
