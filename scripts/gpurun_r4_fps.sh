#!/bin/bash
# round 4: blob-skipping FPS -- parity against the plain kernel on every cloud kind / size, time per round; the oracle tests; the PointNet++ layer
export TMPDIR=/tmp
O=gpurun_out/r4fps; rm -rf $O; mkdir -p $O
timeout 400 python scripts/fps_blob_check.py $O/fps_blob_check.json > $O/check.log 2>&1; grep -c "True" $O/check.log; tail -30 $O/check.log | cut -c1-330
timeout 300 python -m pytest tests/test_primitives_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python scripts/sa_layer_time.py $O/sa_layer.json > $O/sa_layer.log 2>&1; tail -6 $O/sa_layer.log | cut -c1-400
