#!/bin/bash
# usage: scratch/gpu_trace_variant.sh <variant> : kernel trace of 3 bench steps with scratch/variants/<variant>.so -> gpurun_out/tr_<variant>/
set -u
R=$(pwd); O=$R/gpurun_out/tr_$1; mkdir -p $O
cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 > $O/log.txt 2>&1
find $O -name "*.csv" -size +20M -delete; ls $O
