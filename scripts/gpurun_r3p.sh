#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
for i in 1 2; do timeout 300 python scripts/time_predict_batch.py 2> /dev/null | tee -a $O/predict_batch_api.jsonl; done
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; grep real $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value']); [print(e['rng'][:6], e['precision'], e['wall_s'], e['candidates_per_s']) for e in d['api']['predict_batch']]; print(d['cpu_baseline']['cores'], d['cpu_baseline']['value'])"
