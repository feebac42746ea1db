#!/bin/bash
# A/B of two variant builds on one box: usage gpu_ab2.sh <a> <b> ; tile-kernel time averaged over the steps of a run (HIP events), 4 alternations
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; python bench.py --steps 150 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], 'tile(avg) %.4f' % j['roofline']['avg_launch_ms'])"; }
for i in 1 2 3 4; do b $1; b $2; done
