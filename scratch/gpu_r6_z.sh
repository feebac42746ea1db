#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6z; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "file pipeline gz, 1.5 Gbases"; timeout 2400 python scratch/measure_file_pipeline_gz.py 100000 > $O/file_pipeline_big.json 2> $O/file_pipeline_big.err; say "rc $?"; python -c "
import json;j=json.load(open('$O/file_pipeline_big.json'))
print(j['workload'][:120])
for k,v in j['pipeline'].items(): print('%-32s %.4f s  %.3f GB/s text  nodes %d edges %d'%(k,v['seconds'],v['text_gb_per_s'],v['nodes'],v['edges']))"
tail -3 $O/file_pipeline_big.err
say done
