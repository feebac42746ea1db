#!/bin/bash
# kernel timeline of one DEFAULT step (configs[2]): every kernel in order with the gap in front of it
set -u
R=$(pwd); O=$R/gpurun_out/r5k2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o q -- python $R/bench.py --gpus 1 --steps 4 --warmup 2 --cpu-seconds 0 --plain > $O/out.json 2> $O/err.txt
python - $O/kt/q_kernel_trace.csv > $O/timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
sk = [i for i, r in enumerate(rows) if r[2].startswith("void sketch_bs_kernel")]
a, b = sk[-2], sk[-1]          # from the tile launch of the step before the last to the last one: one whole step period
t0 = rows[a][0]
print("step period %.3f ms (tile launch to tile launch)" % ((rows[b][0] - t0) / 1e6))
prev = rows[a][0]; busy = 0
for s, e, n in rows[a:b]:
    print("%8.3f ms  gap %7.3f  dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6 if s > prev else 0.0, (e - s) / 1e6, n))
    busy += max(0, e - max(s, prev)); prev = max(prev, e)
print("busy %.3f ms, idle %.3f ms" % (busy / 1e6, (rows[b][0] - t0 - busy) / 1e6))
PY
cat $O/timeline.txt | head -80
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
