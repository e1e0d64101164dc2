#!/bin/bash
export TMPDIR=/tmp
for t in "$@"; do echo "== $t"; CATGRASP_AMD_LIB=$PWD/build_abl/lib_$t.so timeout 120 python scripts/sa_time.py 2>/dev/null; done
CATGRASP_AMD_LIB=$PWD/build_abl/lib_sa_new.so timeout 300 python -m pytest tests/test_primitives_gpu.py -q -m gpu -k "group or sa_ or abstraction" 2>&1 | tail -3
