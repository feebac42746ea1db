#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_dist_c.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/q_tests.txt | tail -8
bash scratch/gpu_dist_w1.sh
