#!/bin/bash
# usage: scratch/gpu_phase_insts.sh <name> [bench args] : VALU/SALU/LDS instruction counts of the tile kernel cut after phase 1, 2, 3 and complete
set -u
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in 1 2 3 0; do
  MDBG_STOP_PHASE=$P rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/ph$P -o q -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/ph$P.log 2>&1
  python - $O/ph$P/q_counter_collection.csv $O/ph$P/q_kernel_trace.csv $P <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_bs" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[2])) if "sketch_bs" in r["Kernel_Name"]]
print("stop_after", sys.argv[3], "ms", ["%.3f" % d for d in dur], {k: "%.4g" % v for k, v in sorted(agg.items())})
PY
done
find $O -name "*.csv" -size +20M -delete
