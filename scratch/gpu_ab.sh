#!/bin/bash
# A/B of two builds on the same box: usage gpu_ab.sh <variant name> (scratch/variants/<name>.so against the tree's library), tile kernel ms / step ms
cp rust_mdbg_amd/libmdbg_hip.so /tmp/base.so
b() { python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], 'tile %.4f' % j['stage_ms_last_step']['sketch_bs_kernel'], j['stage_ms_last_step'])"; }
for i in 1 2 3; do
  cp /tmp/base.so rust_mdbg_amd/libmdbg_hip.so; b tree
  cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; b $1
done
