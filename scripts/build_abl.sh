#!/bin/bash
# dev only: ablation builds of one source into build_abl/lib_<tag>.so   usage: build_abl.sh <source.hip> <tag> <extra flags...>
set -e
src=$1; tag=$2; shift 2
cd /root/repo
base=$(basename $src .hip)
objs=$(ls catgrasp_amd/csrc/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $src -o build_abl/${base}_$tag.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build_abl/lib_$tag.so $objs build_abl/${base}_$tag.o
