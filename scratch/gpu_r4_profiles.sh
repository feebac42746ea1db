#!/bin/bash
# round 4 profile set: packed + ascii sets (plain runs), stats of the DEFAULT bench command, TCC atomics of the insertion
R=$(pwd); O=$R/gpurun_out/r4p; mkdir -p $O
bash scratch/gpu_profile_set.sh r4p/packed > $O/packed.log 2>&1
bash scratch/gpu_profile_set.sh r4p/ascii --input ascii > $O/ascii.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python $R/bench.py --cpu-seconds 0 > $O/stats_default.json 2> $O/stats_default.err
rocprofv3 -L > $O/counters_avail.txt 2>&1
grep -io "TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\|TCC_REQ[a-z_]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|TCC_EA0_ATOMIC[A-Z0-9_a-z]*" $O/counters_avail.txt | sort -u | head -40 > $O/tcc_names.txt
rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o t -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain > $O/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcc2 -o t -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --plain > $O/tcc2.log 2>&1
cd $R
find $O -name "*.csv" -size +20M -delete
python - <<'PY'
import csv, glob, collections
for sub in ("tcc", "tcc2"):
    for p in glob.glob("gpurun_out/r4p/%s/*counter_collection.csv" % sub):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k.startswith(("insert_windows", "sketch_bs", "fin_mark", "gather")):
                print(sub, k, {c: [round(x) for x in xs[:3]] for c, xs in v.items()})
PY
tail -3 $O/tcc.log $O/tcc2.log; cat $O/tcc_names.txt | head -30
cat $O/packed/summary_kernel_stats.txt | head -14; cut -c1-600 $O/packed/bench.json; echo; cut -c1-400 $O/stats_default.json
