#!/bin/bash
# usage: scratch/gpu_profile_set.sh <name> [bench args]   (run on the GPU box from the repo root)
#   -> gpurun_out/<name>/{stats,fetch,write,sq,sq2,bench.json}: rocprofv3 kernel stats, HBM traffic counters (separate passes),
#      SQ counters (two passes of 8) and the bench line of the same command.  profiles/summarize.py condenses them.
set -u
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --plain "$@" > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain "$@" > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain "$@" > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain "$@" > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain "$@" > $O/sq2.log 2>&1
cd $R
python bench.py --steps 5 --warmup 1 --plain "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" -size +20M -delete
python profiles/summarize.py $O $O/summary && cat $O/summary_kernel_stats.txt | head -12 && cat $O/bench.json
