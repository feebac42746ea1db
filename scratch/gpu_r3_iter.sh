#!/bin/bash
# usage: scratch/gpu_r3_iter.sh <name> [pytest|nopytest] : parity suite, bench line, tile-kernel time cut after each phase (packed input)
set -u
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
if [ "${2:-pytest}" = pytest ]; then timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log; fi
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
for P in 1 2 3; do
  MDBG_STOP_PHASE=$P timeout 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stop_after', $P, 'tile_ms', j['roofline']['avg_launch_ms'])" | tee -a $O/phases.txt
done
