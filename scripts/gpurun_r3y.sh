#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3y; rm -rf $O; mkdir -p $O
for t in "$@"; do
CATGRASP_AMD_LIB=$PWD/build_abl/lib_$t.so timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$t -- python scripts/sa_scaling.py > $O/$t.log 2>&1
find $O/tr_$t -name '*kernel_trace.csv' -exec cp {} $O/ktrace_$t.csv \;
rm -rf $O/tr_$t
python - <<EOF
import csv, collections
rows=[r for r in csv.DictReader(open('$O/ktrace_$t.csv')) if 'sa_reg_kernel' in r['Kernel_Name']]
by=collections.OrderedDict()
for r in rows: by.setdefault(r['Grid_Size']+'/'+r['Workgroup_Size'],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('$t')
for k,v in by.items(): print('  grid',k, 'n',len(v), 'median %.2f us'%sorted(v)[len(v)//2], [round(x,1) for x in v])
EOF
done
