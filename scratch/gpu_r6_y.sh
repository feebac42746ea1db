#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6y; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "gz host"; timeout 1500 bash scratch/gpu_r6_gz.sh > $O/gz.txt 2>&1; say "rc $?"; grep -E "bgzf|best" $O/gz.txt | cut -c1-120
say "file pipeline gz"; timeout 1200 python scratch/measure_file_pipeline_gz.py > $O/file_pipeline.json 2> $O/file_pipeline.err; say "rc $?"; python -c "
import json;j=json.load(open('$O/file_pipeline.json'))
for k,v in j['pipeline'].items(): print('%-32s %.4f s  %.3f GB/s text'%(k,v['seconds'],v['text_gb_per_s']))"
say done
