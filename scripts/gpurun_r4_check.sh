#!/bin/bash
# round 4: the whole GPU suite + smoke + a default-flags bench line on the final tree
export TMPDIR=/tmp
O=gpurun_out/r4check; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err; tail -c 300 $O/bench.err
