#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3v; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_primitives_gpu.py -q -m gpu -k "fps or farthest" 2>&1 | tail -3
timeout 200 python scripts/hbm_kernels.py > $O/hbm.json 2> $O/hbm.err; python -c "
import json; d=json.load(open('$O/hbm.json'))
for r in d['rows'] if 'rows' in d else d:
    if 'farthest' in r.get('kernel',''): print(r['kernel'], r.get('us_per_round'), r.get('us_per_round_all_clouds'), r.get('ms'))
" || tail -5 $O/hbm.err
