export TMPDIR=/tmp
O=gpurun_out/r4suite; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
