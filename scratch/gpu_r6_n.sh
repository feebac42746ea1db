#!/bin/bash
# round 6: the syncmer machine as a scan
set -u
R=$(pwd); O=$R/gpurun_out/r6n; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "syncmer tests"; timeout 1500 python -m pytest tests/test_gpu_syncmers.py -x -q > $O/sync_tests.log 2>&1; say "rc $? $(tail -1 $O/sync_tests.log)"
tail -30 $O/sync_tests.log | cut -c1-200
say "measure"; timeout 600 python scratch/measure_syncmers.py > $O/syncmers.json 2> $O/syncmers.err; say "rc $?"; cat $O/syncmers.json
say done
