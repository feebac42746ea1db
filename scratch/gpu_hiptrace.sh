#!/bin/bash
# host API timeline of one bench step (rocprofv3 --hip-trace): which runtime calls sit between the kernels
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/tr -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 $BENCH_ARGS > $O/trace.log 2>&1
ls $O/tr
python - $O/tr <<'PY' | tee $O/api_timeline.txt
import csv, sys, os
d = sys.argv[1]
api = [r for r in csv.DictReader(open(os.path.join(d, "t_hip_api_trace.csv")))]
ker = [r for r in csv.DictReader(open(os.path.join(d, "t_kernel_trace.csv")))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "API " + r["Function"]) for r in api] + \
     [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  KERNEL " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]) for r in ker]
ev.sort()
sk = [i for i, e in enumerate(ev) if "KERNEL sketch_bs" in e[2]]
a, b = sk[2] - 60, sk[3] + 2          # one whole step: from before the third sketch kernel to the fourth
a = max(a, 0)
t0 = ev[a][0]
for s, e, n in ev[a:b]:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
