#!/bin/bash
# instruction-fetch side of the tile kernel: I-cache requests / hits / misses, fetches, and the wait / busy split of the SQ
R=$(pwd); O=$R/gpurun_out/icache; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/b.log 2>&1
python - $O <<'PY'
import csv, sys, os, collections
for sub in ("a", "b"):
    f = os.path.join(sys.argv[1], sub, "q_counter_collection.csv")
    if not os.path.exists(f): print("missing", f); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sketch_bs_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(sub, k, ["%.4g" % x for x in v[:3]])
PY
