#!/bin/bash
# usage: gpurun_variants_fps.sh "<N:S ...>" <tag> <tag> ...   (build_abl/lib_<tag>.so)
export TMPDIR=/tmp
sizes=$1; shift
for t in "$@"; do echo "== $t"; CATGRASP_AMD_LIB=$PWD/build_abl/lib_$t.so timeout 120 python scripts/fps_time.py $sizes 2>/dev/null; done
