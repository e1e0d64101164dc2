#!/bin/bash
# round 4: blob-skipping FPS -- parity against the plain kernel on every cloud kind / size, time per round, then the oracle tests under each variant
export TMPDIR=/tmp
O=gpurun_out/r4fps; rm -rf $O; mkdir -p $O
timeout 400 python scripts/fps_blob_check.py $O/fps_blob_check.json > $O/check.log 2>&1; tail -45 $O/check.log
for v in blob8 blob4; do
  CATGRASP_AMD_FPS=$v timeout 200 python -m pytest tests/test_primitives_gpu.py -m gpu -x -q -k "farthest or sample_and_group" > $O/pytest_$v.log 2>&1; tail -3 $O/pytest_$v.log
done
for v in blob8 blob4; do
  CATGRASP_AMD_FPS=$v timeout 120 python scripts/sa_layer_time.py $O/sa_layer_$v.json > $O/sa_layer_$v.log 2>&1; tail -3 $O/sa_layer_$v.log | cut -c1-400
done
