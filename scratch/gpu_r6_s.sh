#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6s; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "syncmer tests"; timeout 1500 python -m pytest tests/test_gpu_syncmers.py -x -q > $O/sync_tests.log 2>&1; say "rc $? $(tail -1 $O/sync_tests.log)"
tail -40 $O/sync_tests.log | cut -c1-220
say done
