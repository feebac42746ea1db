#!/bin/bash
# round 6: the WHOLE GPU suite with the new syncmer phase, syncmer throughput, the default bench line
set -u
R=$(pwd); O=$R/gpurun_out/r6r; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
say "syncmers"; timeout 600 python scratch/measure_syncmers.py > $O/syncmers.json 2> $O/syncmers.err; say "rc $?"; cat $O/syncmers.json
say "bench default"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; say "rc $?"; cut -c1-300 $O/bench_default.json
say done
