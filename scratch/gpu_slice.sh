#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/q_tests.txt | tail -6
python bench.py --steps 3 --warmup 1 --cpu-seconds 0 -k 10 --density 0.1 --genome-mb 40 2>gpurun_out/q_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"value\"], j[\"ms_per_step\"], j[\"stage_ms_last_step\"], j[\"graph\"])"
python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>gpurun_out/q_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"value\"], j[\"ms_per_step\"], j[\"stage_ms_last_step\"])"
