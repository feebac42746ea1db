#!/bin/bash
# round 6: one rank of eight incl. the receiver side; file -> .gfa from gzip / BGZF; the decoder alone on this host; then the WHOLE GPU suite
set -u
R=$(pwd); O=$R/gpurun_out/r6g; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "rank w8"; timeout 900 python scratch/measure_rank_w8.py 8 > $O/rank_w8.txt 2>&1; say "rc $?"; tail -5 $O/rank_w8.txt
say "file pipeline gz"; timeout 1200 python scratch/measure_file_pipeline_gz.py > $O/file_pipeline.json 2> $O/file_pipeline.err; say "rc $?"; cat $O/file_pipeline.json | head -60
say "gz host"; timeout 1500 bash scratch/gpu_r6_gz.sh > $O/gz.txt 2>&1; say "rc $?"; head -5 $O/gz.txt; grep -E "threads= (1|8) |threads=32" $O/gz.txt
say "gpu suite"; timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; say "rc $? $(tail -1 $O/gpu_suite.log)"
tail -5 $O/gpu_suite.log
say done
