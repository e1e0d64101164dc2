#!/bin/bash
# the driver's round-end sequence on the shipped tree: GPU suite, smoke, default bench line
export TMPDIR=/tmp
O=gpurun_out/r3check; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; grep real $O/bench.err; head -c 400 $O/bench.json
