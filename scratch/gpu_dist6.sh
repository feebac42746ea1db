#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_dist_c.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/q_tests.txt | tail -4
for v in 150; do echo "MDBG_LISTED_SPAN_MIN=$v"; MDBG_LISTED_SPAN_MIN=$v bash scratch/gpu_dist_kernels.sh 8 segments | head -9; done
