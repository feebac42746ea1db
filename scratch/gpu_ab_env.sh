#!/bin/bash
# A/B of an environment switch on one box: usage gpu_ab_env.sh VAR=1 ; 50-step means, alternating
b() { env "$@" python bench.py --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'step %.4f' % j['ms_per_step'])"; }
for i in 1 2 3 4; do b X=1; b "$@"; done
