#!/bin/bash
# usage: scratch/gpu_parts.sh <out> : step / stage times of the default bench for several MDBG_PARTS settings
set -u
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
for P in "" 86 92 80 "60,90" 100; do
  MDBG_PARTS="$P" timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts=[$P]', 'step_ms %.4f tile_total_ms %.4f launches %d' % (j['ms_per_step'], j['stage_ms_last_step']['sketch_bs_kernel'], j['roofline']['launches_per_step']), j['stage_ms_last_step'])" | tee -a $O/parts.txt
done
