#!/bin/bash
# builds scratch/variants/prio_<abcd>.so: the library with scratch/next_round/tile_setprio.patch applied and -DTILE_PRIO=0x<abcd> (s_setprio value of the tile kernel's phases 1..4),
# plus prio_none.so (patch applied, no priorities: must time like the tree's library).  Then on the GPU box: scratch/gpu_ab_stage.sh 3 prio_none prio_0003 prio_0123 ...
set -e
D=$(cd $(dirname $0)/../.. && pwd)
T=/tmp/csrc_prio; rm -rf $T; mkdir -p $T/rust_mdbg_amd $D/scratch/variants
cp -r $D/rust_mdbg_amd/csrc $T/rust_mdbg_amd/; cp -r $D/include $T/
(cd $T && patch -p1 < $D/scratch/next_round/tile_setprio.patch)
cd $T/rust_mdbg_amd/csrc
[ -f edges.o ] || make edges.o
for v in none "$@"; do
  F=""; [ "$v" != none ] && F="-DTILE_PRIO=0x$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $F -c -o /tmp/prio_$v.o libmdbg.hip &
done
wait
for v in none "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $D/scratch/variants/prio_$v.so /tmp/prio_$v.o edges.o -ldl; done
ls -la $D/scratch/variants/prio_*.so
