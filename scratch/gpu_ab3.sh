#!/bin/bash
# several variant builds on one box: bench step / averaged tile kernel time (150 steps), alternating, 3 rounds; the bench checks the graph of every run
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; python bench.py --steps 150 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], 'tile(avg) %.4f' % j['roofline']['avg_launch_ms'], j['graph']['nodes'])"; }
for i in 1 2 3; do for v in "$@"; do b $v; done; done
