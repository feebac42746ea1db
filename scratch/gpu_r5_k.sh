#!/bin/bash
# kernel timeline of one human step: where is the GPU idle?
set -u
R=$(pwd); O=$R/gpurun_out/r5k${TAG:-}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env ${EXTRA:-X=1} rocprofv3 --kernel-trace --output-format csv -d $O/kt -o q -- python $R/bench.py --gpus 1 --workload human --steps 2 --warmup 1 --cpu-seconds 0 --plain > $O/out.json 2> $O/err.txt
python - $O/kt/q_kernel_trace.csv > $O/timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last step = from the last clear_table-ish gap: take the last 1/3 of the sketch_bs launches
sk = [i for i, r in enumerate(rows) if r[2].startswith("void sketch_bs_kernel")]
first = sk[-8]            # the last 8 tile launches = the last step
i0 = first
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 2_000_000 and not rows[i0 - 1][2].startswith("fin_emit"): i0 -= 1
seg = rows[i0:]
t0 = seg[0][0]; busy = 0; gaps = []
prev_end = seg[0][0]
for s, e, n in seg:
    if s > prev_end: gaps.append((s - prev_end, n, (s - t0) / 1e6))
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
span = prev_end - t0
print("last step: span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
for g, n, at in sorted(gaps, reverse=True)[:25]: print("gap %.3f ms before %s at %.2f ms" % (g / 1e6, n, at))
import collections
c = collections.Counter()
for s, e, n in seg: c[n] += e - s
for n, t in c.most_common(25): print("%-42s %.3f ms" % (n, t / 1e6))
print("the step's first kernels (start, end in ms from the step's start):")
for s, e, n in seg[:14]: print("  %8.3f %8.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n))
PY
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -delete
