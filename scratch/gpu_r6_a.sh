#!/bin/bash
# round 6, first call: where the round starts on this box — default line, human N=1, human through the multi-GPU layer at one rank (with the layer's own
# stage timers), and the kernel timelines of a default step and of a dist step (what the launches between the big kernels are, what finalize is made of)
set -u
R=$(pwd); O=$R/gpurun_out/r6a; mkdir -p $O
python bench.py --cpu-seconds 0 --no-scale-anchor > $O/default.json 2> $O/default.err
python bench.py --gpus 1 --workload human --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err
MDBG_DIST_TIMING=1 python bench.py --gpus 1 --workload human --force-dist --cpu-seconds 0 --steps 10 > $O/human_dist_w1.json 2> $O/human_dist_w1.err
python - <<PY
import json
for f in ('default', 'human_n1', 'human_dist_w1'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value'], 1), round(j['ms_per_step'], 3), j['stage_ms_last_step'], (j.get('no_exchange_anchor') or {}).get('ms_per_step'))
    except Exception as e: print(f, 'failed', e)
PY
grep "dist timing" $O/human_dist_w1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -o q -- python $R/bench.py --gpus 1 --steps 4 --warmup 2 --cpu-seconds 0 --plain > $O/kt_def.json 2> $O/kt_def.err
MODE=def python $R/scratch/timeline.py $O/kt_def/q_kernel_trace.csv > $O/timeline_default.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dist -o q -- python $R/bench.py --gpus 1 --workload human --force-dist --steps 2 --warmup 1 --cpu-seconds 0 --plain > $O/kt_dist.json 2> $O/kt_dist.err
MODE=human python $R/scratch/timeline.py $O/kt_dist/q_kernel_trace.csv > $O/timeline_dist.txt
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
head -70 $O/timeline_default.txt; head -90 $O/timeline_dist.txt
