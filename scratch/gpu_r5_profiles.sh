#!/bin/bash
# round 5 profile set: packed set (kernel stats, FETCH / WRITE, two SQ passes, bench --plain), stats of the DEFAULT bench command, the default line, the human line,
# the one-rank cost of the multi-GPU layer with its stage timers
R=$(pwd); O=$R/gpurun_out/r5p; mkdir -p $O
bash scratch/gpu_profile_set.sh r5p/packed > $O/packed.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python $R/bench.py --cpu-seconds 0 > $O/stats_default.json 2> $O/stats_default.err
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --workload human --steps 8 --warmup 2 --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err
for ch in 1 2; do MDBG_DIST_TIMING=1 python bench.py --gpus 1 --force-dist --workload human --steps 8 --warmup 2 --cpu-seconds 0 --chunks $ch > $O/dist_w1_c$ch.json 2> $O/dist_w1_c$ch.err; done
python bench.py --gpus 1 --workload human --multik --steps 3 --warmup 1 --cpu-seconds 0 > $O/multik_n1.json 2> $O/multik_n1.err
find $O -name "*.csv" -size +20M -delete; find $O -name "*.db" -delete
cat $O/packed/summary_kernel_stats.txt | head -14; grep "dist timing" $O/*.err
