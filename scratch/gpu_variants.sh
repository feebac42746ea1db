#!/bin/bash
# usage: scratch/gpu_variants.sh <out> <variant> ... : tile-kernel time of each scratch/variants/<variant>.so on the default bench ("base" = the tree's library)
set -u
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O
cp rust_mdbg_amd/libmdbg_hip.so /tmp/base.so
for V in "$@"; do
  if [ $V = base ]; then cp /tmp/base.so rust_mdbg_amd/libmdbg_hip.so; else cp scratch/variants/$V.so rust_mdbg_amd/libmdbg_hip.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', 'tile_ms %.4f step_ms %.4f' % (j['roofline']['avg_launch_ms'], j['ms_per_step']), j['stage_ms_last_step'])" | tee -a $O/variants.txt
done
cp /tmp/base.so rust_mdbg_amd/libmdbg_hip.so
