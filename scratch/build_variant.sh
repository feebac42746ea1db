#!/bin/bash
# usage: scratch/build_variant.sh <name> "<extra hipcc flags>"  -> scratch/variants/<name>.so (a libmdbg_hip.so built with the flags)
# On the GPU box: cp scratch/variants/<name>.so rust_mdbg_amd/libmdbg_hip.so  (the box's copy of the repo is scratch)
set -e
N=$1; shift
D=$(cd $(dirname $0)/.. && pwd)
mkdir -p $D/scratch/variants /tmp/variant_$N
cd $D/rust_mdbg_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c -o /tmp/variant_$N/libmdbg.o libmdbg.hip
[ -f edges.o ] || make edges.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $D/scratch/variants/$N.so /tmp/variant_$N/libmdbg.o edges.o -ldl
ls -la $D/scratch/variants/$N.so
