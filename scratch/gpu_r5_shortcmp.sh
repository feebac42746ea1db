#!/bin/bash
# EXPERIMENT: how much of the insertion is the representative's traffic? (variant compares 16 bytes of the representative instead of 8k)
cd /root/repo; mkdir -p gpurun_out/r5x
bash scratch/gpu_ab_stage.sh 2 base shortcmp > gpurun_out/r5x/shortcmp_fly.txt 2>&1; cat gpurun_out/r5x/shortcmp_fly.txt
h() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; python bench.py --workload human --steps 3 --warmup 1 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'human step %.3f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
(h base; h shortcmp; h base; h shortcmp) > gpurun_out/r5x/shortcmp_human.txt 2>&1; cat gpurun_out/r5x/shortcmp_human.txt
