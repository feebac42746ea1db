#!/bin/bash
# round 6, third call: the launch consolidation (block sums by the tiles, count + reserve in one launch, finalize counters left clean, batch table uploaded once, the clear on
# the main stream with the small counters folded in, events by level): parity first, then the default line with / without the stage events in the timed region, and the timeline
set -u
R=$(pwd); O=$R/gpurun_out/r6c; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8
for i in 1 2; do
python bench.py --cpu-seconds 0 --no-scale-anchor > $O/default_$i.json 2> $O/default_$i.err
MDBG_BENCH_STAGE_EVENTS=1 python bench.py --cpu-seconds 0 --no-scale-anchor > $O/default_ev_$i.json 2> $O/default_ev_$i.err
done
python bench.py --gpus 1 --workload human --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err
python - <<PY
import json
for f in ('default_1', 'default_ev_1', 'default_2', 'default_ev_2', 'human_n1'):
    try:
        j = json.load(open('$O/%s.json' % f)); s = j['stage_ms_last_step']
        print(f, round(j['value'], 1), round(j['ms_per_step'], 4), {k: round(v, 4) for k, v in s.items() if k != 'measured_in'}, 'residual %.4f' % (j['ms_per_step'] - s['sketch'] - s['insert'] - s['finalize']), j['graph'].get('node_digest'))
    except Exception as e: print(f, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -o q -- python $R/bench.py --gpus 1 --steps 4 --warmup 2 --cpu-seconds 0 --plain > $O/kt_def.json 2> $O/kt_def.err
MODE=def python $R/scratch/timeline.py $O/kt_def/q_kernel_trace.csv > $O/timeline_default.txt
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
head -40 $O/timeline_default.txt
