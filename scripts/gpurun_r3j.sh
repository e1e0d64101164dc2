#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3j; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "sa_reg_kernel" --output-format csv -d $O/kt -- python scripts/sa_scaling.py > $O/kt.log 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $O/sa_scaling.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'sa_reg' in r['Kernel_Name']]
sizes = (1, 2, 4, 8, 16, 24, 32, 48)
for i, B in enumerate(sizes):
    v = d[6 * i + 2: 6 * i + 6]
    print(B, 'clouds', round(sum(v) / len(v) / 1000, 2), 'us', [round(x / 1000, 1) for x in d[6 * i: 6 * i + 6]])
PY
rm -rf $O/kt
