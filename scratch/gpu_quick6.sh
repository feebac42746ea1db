#!/bin/bash
b() { python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>gpurun_out/q_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_last_step']; print('$1 step_ms %.4f gaps %.4f' % (j['ms_per_step'], j['ms_per_step']-s['sketch']-s['insert']-s['finalize']), s, j['value'])"; }
b a; b b
timeout 1700 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_config34.py > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/q_tests.txt | tail -3
