#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_dist_c.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/q_tests.txt | tail -8
rm -f gpurun_out/dist_traffic.jsonl
bash scratch/gpu_dist_traffic.sh "2 2" "4 2" "8 2" "4 3"
