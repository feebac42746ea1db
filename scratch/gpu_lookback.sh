#!/bin/bash
# look-back experiment: running count of the tile kernel computed in-kernel vs the scan kernels (MDBG_LOOKBACK), cost on the tile kernel
run() { timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>gpurun_out/lb_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step_ms %.4f' % j['ms_per_step'], j['stage_ms_last_step'])"; grep "look-back" gpurun_out/lb_err.txt | tail -2; }
run base
MDBG_LOOKBACK=1 MDBG_LOOKBACK_CHECK=1 run lookback_check
MDBG_LOOKBACK=1 run lookback
run base
MDBG_LOOKBACK=1 run lookback
