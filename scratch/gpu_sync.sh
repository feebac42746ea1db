#!/bin/bash
python scratch/measure_syncmers.py 2>/dev/null | tail -1 | python -c "
import json,sys
for r in json.loads(sys.stdin.read()): print(r['l'], r['s'], r['density'], 'tile kernel %.3f ms  %.1f Gbases/s' % (r['ms_kernel'], r['gbases_per_s']), r['minimizers'])"
timeout 900 python -m pytest tests/test_gpu_syncmers.py -x -q -m gpu > gpurun_out/q_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/q_tests.txt | tail -3
