#!/bin/bash
export TMPDIR=/tmp
for t in "$@"; do echo "== $t"; CATGRASP_AMD_LIB=$PWD/build_abl/lib_$t.so timeout 120 python scripts/sa_time.py 2>/dev/null; done
