#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6w; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "dry run tests"; timeout 1500 python -m pytest tests/test_gpu_bench_dry_run.py -x -q > $O/dry.log 2>&1; say "rc $? $(tail -1 $O/dry.log)"
tail -25 $O/dry.log | cut -c1-250
say done
