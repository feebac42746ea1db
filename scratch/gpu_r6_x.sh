#!/bin/bash
# round 6: the per-entry insertion with the wave's hashes staged in LDS (MDBG_LISTED_STAGE=1) against without
set -u
R=$(pwd); O=$R/gpurun_out/r6x; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
for st in 0 1 0 1; do say "rank w8 stage $st"; MDBG_LISTED_STAGE=$st timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_stage$st.txt 2>&1; say "rc $?"; grep -E "^receiver pass 2" $O/rank_w8_stage$st.txt | cut -c1-200; done
say "dist tests staged"; MDBG_LISTED_STAGE=1 timeout 1800 python -m pytest tests -m gpu -x -q -k "dist or rank or partition or listed or owner" > $O/gpu_dist.log 2>&1; say "rc $? $(tail -1 $O/gpu_dist.log)"
say done
