#!/bin/bash
# usage: scratch/gpu_sq_counters.sh <name> [bench args]  (GPU box, repo root) -> gpurun_out/<name>/{sq,sq2}: SQ counters of one bench step
# (two passes of 8 SQ counters each; --kernel-trace only, as the pool requires)
set -u
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > $O/sq2.log 2>&1
cd $R
python - "$O" <<'PY'
import csv, sys, json, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for sub in ("sq", "sq2"):
    try:
        for r in csv.DictReader(open("%s/%s/q_counter_collection.csv" % (O, sub))):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    except Exception as e:
        print("missing", sub, e)
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items() if "sketch" in k or "insert_windows" in k or "fin_" in k or "gather" in k}
json.dump(out, open(O + "/sq_summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: "%.4g" % v for c, v in d.items()})
PY
find $O -name "*.csv" -size +20M -delete
