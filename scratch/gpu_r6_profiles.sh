#!/bin/bash
# round 6 profile set: packed set (kernel stats, FETCH / WRITE, two SQ passes, bench --plain), per-phase SQ counters (stop points 1 2 3 4 5 and the whole kernel), stats of the
# DEFAULT bench command, the default line, the human line, the multi-GPU layer at one rank, the multik sweep.  Every command under its own timeout.
set -u
R=$(pwd); O=$R/gpurun_out/r6p; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "profile set"; timeout 1200 bash scratch/gpu_profile_set.sh r6p/packed > $O/packed.log 2>&1; say "rc $?"
cd /tmp && export TMPDIR=/tmp
: > $O/phase_counters.txt
for P in 1 2 3 4 5 0; do
  say "phase $P"
  MDBG_STOP_PHASE=$P timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/ph$P -o q -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --plain > $O/ph$P.log 2>&1
  python - $O/ph$P/q_counter_collection.csv $O/ph$P/q_kernel_trace.csv $P >> $O/phase_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_bs" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[2])) if "sketch_bs" in r["Kernel_Name"]]
print("stop_after", sys.argv[3], "launches", len(dur), "ms", ["%.3f" % d for d in dur], {k: "%.6g" % v for k, v in sorted(agg.items())})
PY
done
say "stats of the default command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python $R/bench.py --cpu-seconds 0 > $O/stats_default.json 2> $O/stats_default.err; say "rc $?"
cd $R
say "default line"; timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; say "rc $?"
say "human"; timeout 600 python bench.py --gpus 1 --workload human --steps 8 --warmup 2 --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err; say "rc $?"
say "dist w1"; MDBG_DIST_TIMING=1 timeout 600 python bench.py --gpus 1 --force-dist --workload human --steps 8 --warmup 2 --cpu-seconds 0 > $O/dist_w1.json 2> $O/dist_w1.err; say "rc $?"
say "multik"; timeout 900 python bench.py --gpus 1 --workload human --multik --steps 3 --warmup 1 --cpu-seconds 0 > $O/multik_n1.json 2> $O/multik_n1.err; say "rc $?"
find $O -name "*.csv" -size +8M -delete; find $O -name "*.db" -delete
cat $O/packed/summary_kernel_stats.txt | head -16; cat $O/phase_counters.txt | cut -c1-200; grep "dist timing" $O/*.err
python - <<PY
import json
for f in ('bench_default', 'human_n1', 'dist_w1', 'multik_n1'):
    try:
        j = json.load(open('$O/%s.json' % f)); s = j['stage_ms_last_step']
        print(f, round(j['value'], 1), round(j['ms_per_step'], 4), {k: round(v, 4) for k, v in s.items() if k != 'measured_in'}, 'frac', j['roofline']['frac'], (j.get('no_exchange_anchor') or {}).get('ms_per_step'))
        if j.get('cpu_baseline'): print('   cpu', {k: j['cpu_baseline'][k] for k in ('value', 'cores', 'nodes', 'windows', 'node_digest', 'whole_workload', 'matches_gpu')})
        if j.get('scale_anchor_n1'): print('   anchor', j['scale_anchor_n1'].get('value'), j['scale_anchor_n1'].get('shard_vs_oracle'))
        if j.get('roofline_hpc_input'): print('   hpc', j['roofline_hpc_input']['avg_launch_ms'], j['roofline_hpc_input']['frac'])
    except Exception as e: print(f, 'failed', e)
PY
say done
