#!/bin/bash
# usage: scratch/gpu_trace.sh <out> [env assignments]: kernel timeline (start / end, us) of the last bench step
set -u
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/trace.log 2>&1
python - $O/tr/t_kernel_trace.csv <<'PY' | tee $O/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last packed step: find the last 'clear_table' ... take the last 40 kernels before the ASCII runs: simpler: print the window around the 3rd-from-last sketch_bs launch group
idx = [i for i, r in enumerate(rows) if "sketch_bs" in r["Kernel_Name"]]
# steps: warmup, 2 steps (packed), then 2 ascii sketches, then 1 packed ingest
sel = idx[-1]
lo = max(0, sel - 14); hi = min(len(rows), sel + 22)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print("%-40s q%-3s %9.1f -> %9.1f  (%7.1f us)" % (r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
