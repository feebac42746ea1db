#!/bin/bash
# round 6: SQ counters of the syncmer kernel (l = 12, s = 4) up to the end of phase 2 / 3 / 4
set -u
R=$(pwd); O=$R/gpurun_out/r6q; mkdir -p $O
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
cd /tmp && export TMPDIR=/tmp
: > $O/sync_phase_counters.txt
for S in 4; do for P in 2 3 4 0; do
  say "s $S phase $P"
  MDBG_STOP_PHASE=$P timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/s${S}ph$P -o q -- python $R/scratch/measure_syncmers_one.py 12 $S 0.05 > $O/s${S}ph$P.log 2>&1
  python - $O/s${S}ph$P/q_counter_collection.csv $O/s${S}ph$P/q_kernel_trace.csv "s=$S stop=$P" >> $O/sync_phase_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sketch_bs" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
for r in rows:
    if int(r["Dispatch_Id"]) == last: agg[r["Counter_Name"]] += float(r["Counter_Value"])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[2])) if "sketch_bs" in r["Kernel_Name"]]
print(sys.argv[3], "ms", ["%.3f" % d for d in dur], {k: "%.6g" % v for k, v in sorted(agg.items())})
PY
done; done
for S in 4; do for P in 2 3 4 0; do
  say "wait counters s $S phase $P"
  MDBG_STOP_PHASE=$P timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/w${S}ph$P -o q -- python $R/scratch/measure_syncmers_one.py 12 $S 0.05 > $O/w${S}ph$P.log 2>&1
  python - $O/w${S}ph$P/q_counter_collection.csv $O/w${S}ph$P/q_kernel_trace.csv "s=$S stop=$P" >> $O/sync_phase_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sketch_bs" in r["Kernel_Name"]]
if rows:
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[3], {k: "%.6g" % v for k, v in sorted(agg.items())})
PY
done; done
find $O -name "*.csv" -size +8M -delete; find $O -name "*.db" -delete
cat $O/sync_phase_counters.txt
say done
