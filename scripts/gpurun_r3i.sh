#!/bin/bash
# round 3, call I: the multi-rank control flow of the new bench defaults on ONE device (gloo, dev check) + the full GPU suite + smoke
export TMPDIR=/tmp
O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
CATGRASP_BENCH_BACKEND=gloo CATGRASP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --secondary "" > $O/bench_2rank_dev.json 2> $O/bench_2rank_dev.err; tail -c 600 $O/bench_2rank_dev.err; head -c 1500 $O/bench_2rank_dev.json; echo
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
