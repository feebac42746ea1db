#!/bin/bash
# round 6: clean A/B of the listed insertion (the copies of the peers' sketches finished before the clock starts), the segments' pack / scatter (old and new copy kernel), GPU suite
set -u
R=$(pwd); O=$R/gpurun_out/r6j; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "rank w8 default"; timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_default.txt 2>&1; say "rc $?"; grep -E "^receiver|^ms per|^segments" $O/rank_w8_default.txt | cut -c1-400
say "rank w8 old"; MDBG_SEG_OLD=1 MDBG_LISTED_OLD=1 timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_old.txt 2>&1; say "rc $?"; grep -E "^receiver|^segments" $O/rank_w8_old.txt | cut -c1-400
say "rank w8 span"; MDBG_LISTED_SPAN_MIN=0 timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_span.txt 2>&1; say "rc $?"; grep -E "^receiver" $O/rank_w8_span.txt | cut -c1-400
say "rank w8 span100"; MDBG_LISTED_SPAN_MIN=100 timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_span100.txt 2>&1; say "rc $?"; grep -E "^receiver" $O/rank_w8_span100.txt | cut -c1-400
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
say done
