#!/bin/bash
# round 6: segments with fused counts (no per-entry count / prefix arrays), the WHOLE GPU suite, the multi-GPU layer at one rank
set -u
R=$(pwd); O=$R/gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
say "rank w8"; timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8.txt 2>&1; say "rc $?"; grep -E "^receiver pass 2|^ms per|^segments" $O/rank_w8.txt | cut -c1-400
say "trace"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/scratch/measure_rank_w8.py 8 > $O/trace.log 2>&1); say "rc $?"
t=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && { head -1 "$t" > $O/trace_tail.csv; tail -400 "$t" >> $O/trace_tail.csv; rm -f "$t"; }
rm -f $O/trace/*/*agent_info.csv
say "dist w1 human"; timeout 900 python bench.py --gpus 1 --force-dist --workload human > $O/dist_w1.json 2> $O/dist_w1.err; say "rc $?"; cut -c1-300 $O/dist_w1.json
say done
