#!/bin/bash
# bench line + the quick half of the GPU suite
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>gpurun_out/q_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms %.4f' % j['ms_per_step'], j['stage_ms_last_step'], j['value'])"
tail -3 gpurun_out/q_err.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_fuzz.py tests/test_gpu_syncmers.py tests/test_gpu_lmer.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5
