#!/bin/bash
# round 6: the per-entry kernel with fewer round trips, at 5 / 6 / 8 waves per SIMD
set -u
R=$(pwd); O=$R/gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
for w in 5 6 8; do
say "rank w8 waves $w"; MDBG_LISTED_WAVES=$w timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_waves$w.txt 2>&1; say "rc $?"; grep -E "^receiver pass 2" $O/rank_w8_waves$w.txt | cut -c1-200
done
say "gpu dist tests"; timeout 1800 python -m pytest tests -m gpu -x -q -k "dist or rank or partition or listed or owner" > $O/gpu_dist.log 2>&1; say "rc $? $(tail -1 $O/gpu_dist.log)"
say done
