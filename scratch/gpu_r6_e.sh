#!/bin/bash
# after the call that never came back: every command under its own timeout, progress lines into gpurun_out/r6e/progress.txt
set -u
R=$(pwd); O=$R/gpurun_out/r6e; mkdir -p $O
P=$O/progress.txt; : > $P
say() { echo "$(date +%T) $*" >> $P; echo "$(date +%T) $*"; }
say start; rocm-smi --showuse 2>/dev/null | head -8 >> $P
say smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; say "rc $? $(tail -1 $O/smoke.txt | cut -c1-200)"
say "no fuse"; MDBG_NO_FUSE=1 timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain --steps 3 > $O/nofuse.json 2> $O/nofuse.err; say "rc $? $(grep -m1 -i fault $O/nofuse.err)"
say "no fuse no tile sums"; MDBG_NO_TILE_SUMS=1 MDBG_NO_FUSE=1 timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain --steps 3 > $O/nofuse_nosums.json 2> $O/nofuse_nosums.err; say "rc $? $(grep -m1 -i fault $O/nofuse_nosums.err)"
say "small genome"; timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain --steps 3 --genome-mb 20 > $O/small.json 2> $O/small.err; say "rc $? $(grep -m1 -i fault $O/small.err)"
say "default plain"; timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/a_1.json 2> $O/a_1.err; say "rc $?"
say "no tile sums"; MDBG_NO_TILE_SUMS=1 timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/b_1.json 2> $O/b_1.err; say "rc $?"
say "default plain 2"; timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/a_2.json 2> $O/a_2.err; say "rc $?"
say "no tile sums 2"; MDBG_NO_TILE_SUMS=1 timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain > $O/b_2.json 2> $O/b_2.err; say "rc $?"
python - <<PY
import json
for f in ('a_1', 'b_1', 'a_2', 'b_2'):
    try:
        j = json.load(open('$O/%s.json' % f)); s = j['stage_ms_last_step']
        print(f, round(j['value'], 1), round(j['ms_per_step'], 4), 'tile %.4f' % j['roofline']['avg_launch_ms'], {k: round(v, 4) for k, v in s.items() if k != 'measured_in'})
    except Exception as e: print(f, 'failed', e)
PY
for t in tests/test_gpu_syncmers.py tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py tests/test_gpu_dist_c.py tests/test_gpu_distributed.py tests/test_gpu_dist_scale.py; do
  say "pytest $t"; timeout 900 python -m pytest $t -x -q > $O/$(basename $t .py).log 2>&1; say "rc $? $(tail -1 $O/$(basename $t .py).log)"
done
say "bench dry run tests"; timeout 1500 python -m pytest tests/test_gpu_bench_dry_run.py -x -q > $O/bench_dry.log 2>&1; say "rc $? $(tail -1 $O/bench_dry.log)"
say "human dist w1"; MDBG_DIST_TIMING=1 timeout 600 python bench.py --gpus 1 --workload human --force-dist --cpu-seconds 0 --steps 10 > $O/human_dist_w1.json 2> $O/human_dist_w1.err; say "rc $?"
grep "dist timing" $O/human_dist_w1.err
python - <<PY
import json
try:
    j = json.load(open('$O/human_dist_w1.json')); print('human_dist_w1', round(j['value'], 1), round(j['ms_per_step'], 3), j['stage_ms_last_step'], (j.get('no_exchange_anchor') or {}).get('ms_per_step'), j['graph'].get('node_digest'))
except Exception as e: print('human_dist_w1 failed', e)
PY
say "rank w8"; timeout 600 python scratch/measure_rank_w8.py 8 > $O/rank_w8.txt 2>&1; say "rc $?"; cat $O/rank_w8.txt | tail -4
say "dist timeline"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt_dist -o q -- python $R/bench.py --gpus 1 --workload human --force-dist --steps 2 --warmup 1 --cpu-seconds 0 --plain > $O/kt_dist.json 2> $O/kt_dist.err; say "rc $?"
MODE=dist timeout 300 python $R/scratch/timeline.py $O/kt_dist/q_kernel_trace.csv > $O/timeline_dist.txt 2>&1
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
head -120 $O/timeline_dist.txt
say done
