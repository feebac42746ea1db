#!/bin/bash
R=$(pwd); O=$R/gpurun_out/ldsc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INST_LEVEL_LDS\|SQ_WAIT_INST_LDS" | sort -u | tr '\n' ' '; echo
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $O/a -o q -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $O/a.log 2>&1
python - $O <<'PY'
import csv, sys, os, collections
f = os.path.join(sys.argv[1], "a", "q_counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "sketch_bs_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k, ["%.4g" % x for x in v[:2]])
PY
