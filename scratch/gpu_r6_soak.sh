#!/bin/bash
# round 6: the multi-rank and syncmer tests five times over (every other pass with poisoned allocations): rare races in this round's kernels
set -u
R=$(pwd); O=$R/gpurun_out/r6soak; mkdir -p $O
for i in 1 2 3 4 5; do
  if [ $((i % 2)) = 0 ]; then export MDBG_POISON=1; else unset MDBG_POISON; fi
  timeout 1200 python -m pytest tests/test_gpu_dist_scale.py tests/test_gpu_dist_c.py tests/test_gpu_syncmers.py tests/test_gpu_round6.py tests/test_gpu_bench_dry_run.py -x -q > $O/pass$i.log 2>&1; echo "pass $i rc $? $(tail -1 $O/pass$i.log)"
done
