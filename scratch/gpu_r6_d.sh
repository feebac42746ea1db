#!/bin/bash
# A/B on one box, alternating: the tiles add up the scan's first level themselves (default) vs MDBG_NO_TILE_SUMS; then parity of the sketch paths, then the dist timeline
set -u
R=$(pwd); O=$R/gpurun_out/r6d; mkdir -p $O
for i in 1 2 3; do
python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/a_$i.json 2> $O/a_$i.err
MDBG_NO_TILE_SUMS=1 python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/b_$i.json 2> $O/b_$i.err
done
python - <<PY
import json
for f in ('a_1', 'b_1', 'a_2', 'b_2', 'a_3', 'b_3'):
    try:
        j = json.load(open('$O/%s.json' % f)); s = j['stage_ms_last_step']
        print(f, round(j['value'], 1), round(j['ms_per_step'], 4), 'tile %.4f' % j['roofline']['avg_launch_ms'], {k: round(v, 4) for k, v in s.items() if k != 'measured_in'})
    except Exception as e: print(f, 'failed', e)
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py tests/test_gpu_syncmers.py -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dist -o q -- python $R/bench.py --gpus 1 --workload human --force-dist --steps 2 --warmup 1 --cpu-seconds 0 --plain > $O/kt_dist.json 2> $O/kt_dist.err
MODE=dist python $R/scratch/timeline.py $O/kt_dist/q_kernel_trace.csv > $O/timeline_dist.txt
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
head -150 $O/timeline_dist.txt
