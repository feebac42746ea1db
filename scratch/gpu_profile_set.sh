#!/bin/bash
# usage: scratch/gpu_profile_set.sh <name>   (run on the GPU box from the repo root) -> gpurun_out/<name>/{stats,fetch,write,bench.json}
set -u
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/write.log 2>&1
cd $R
python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" -size +20M -delete
