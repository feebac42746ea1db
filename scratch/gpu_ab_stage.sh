#!/bin/bash
# variants alternating, N rounds (first arg), step + stage times shown
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; python bench.py --steps 150 --warmup 5 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], 'tile(avg) %.4f' % j['roofline']['avg_launch_ms'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
N=$1; shift
for i in $(seq $N); do for v in "$@"; do b $v; done; done
