#!/bin/bash
# round 6: where the span kernel beats the per-entry kernel now (2 and 4 ranks), the segment passes on 4,096-entry blocks, the GPU suite with the stage timers, dist dry runs
set -u
R=$(pwd); O=$R/gpurun_out/r6m; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
for w in 2 4; do for sm in 0 1000000; do
say "rank w$w span_min $sm"; MDBG_LISTED_SPAN_MIN=$sm timeout 900 python scratch/measure_rank_w8.py $w > $O/rank_w${w}_sm$sm.txt 2>&1; say "rc $?"; grep -E "^receiver pass 2|^segments" $O/rank_w${w}_sm$sm.txt | cut -c1-300
done; done
say "rank w8"; timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8.txt 2>&1; say "rc $?"; grep -E "^receiver pass 2|^ms per|^segments" $O/rank_w8.txt | cut -c1-400
say "dist w1 human"; timeout 900 python bench.py --gpus 1 --force-dist --workload human > $O/dist_w1.json 2> $O/dist_w1.err; say "rc $?"; cut -c1-300 $O/dist_w1.json
say done
