#!/bin/bash
# EXPERIMENT: the table's atomics at workgroup scope (L2 atomics; WRONG results across XCDs) — how much of the insertion is the memory-side atomic rate?
cd /root/repo; mkdir -p gpurun_out/r5w
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; MDBG_STOP_PHASE=0 python bench.py --steps 50 --warmup 5 --cpu-seconds 0 --plain 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'], j['graph']['distinct'])"; }
(b cur; b wgatom; b cur; b wgatom) > gpurun_out/r5w/wg_atomics.txt 2>&1; cat gpurun_out/r5w/wg_atomics.txt
