#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6t; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "dist scale tests"; timeout 1500 python -m pytest tests/test_gpu_dist_scale.py -x -q > $O/dist_scale.log 2>&1; say "rc $? $(tail -1 $O/dist_scale.log)"
tail -30 $O/dist_scale.log | cut -c1-220
say done
