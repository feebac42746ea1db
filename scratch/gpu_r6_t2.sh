#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6t2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t.log 2>&1; echo "rc $? $(tail -1 $O/t.log)"; tail -25 $O/t.log | cut -c1-220
