#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
for n in 50000 25000 12500 6250; do
timeout 300 python bench.py --candidates $n --steps 10 --warmup 3 --secondary "" --no-cpu-baseline --no-api --no-pmc-traffic --no-rccl-selftest > $O/b_$n.json 2> $O/b_$n.err
python -c "
import json; d=json.load(open('$O/b_$n.json')); print($n, d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
