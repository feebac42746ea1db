#!/bin/bash
# round 6: the receiver side of one rank of eight under three insertion variants (per-entry, span kernel, per-entry through upsert_wave), a kernel trace of the default,
# the WHOLE GPU suite, the default bench line
set -u
R=$(pwd); O=$R/gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "rank w8 default"; timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_default.txt 2>&1; say "rc $?"; grep -E "^receiver|^ms per" $O/rank_w8_default.txt | cut -c1-330
say "rank w8 span"; MDBG_LISTED_SPAN_MIN=0 timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_span.txt 2>&1; say "rc $?"; grep -E "^receiver" $O/rank_w8_span.txt | cut -c1-330
say "rank w8 wave"; MDBG_LISTED_WAVE=1 timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8_wave.txt 2>&1; say "rc $?"; grep -E "^receiver" $O/rank_w8_wave.txt | cut -c1-330
say "trace"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/scratch/measure_rank_w8.py 8 > $O/trace.log 2>&1); say "rc $?"
f=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200
# keep the trace small: the stats only + the last 3000 rows of the kernel trace
t=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && { head -1 "$t" > $O/trace_tail.csv; tail -3000 "$t" >> $O/trace_tail.csv; rm -f "$t"; }
rm -f $O/trace/*/*agent_info.csv
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
say "bench default"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; say "rc $?"; cut -c1-600 $O/bench_default.json
say done
