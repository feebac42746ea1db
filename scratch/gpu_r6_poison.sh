#!/bin/bash
# round 6: the WHOLE GPU suite with every device block handed out filled with 0xA5 (MDBG_POISON): nothing of this round's kernels may lean on a fresh allocation's zeros
set -u
R=$(pwd); O=$R/gpurun_out/r6poison; mkdir -p $O
MDBG_POISON=1 timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite_poison.log 2>&1; echo "rc $? $(tail -1 $O/gpu_suite_poison.log)"; tail -12 $O/gpu_suite_poison.log | cut -c1-220
