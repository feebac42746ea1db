#!/bin/bash
# usage: gpu_dist_kernels.sh <W> <mode: segments | whole> : GPU kernel time per rank and step of the C layer with W thread-ranks on one GPU (ranks run
# concurrently, so single kernels are slowed by their neighbours: an upper bound of the per-rank work), 3 steps per rank
R=$(pwd); O=$R/gpurun_out/distk_$1_$2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODES=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/scratch/measure_dist_traffic.py $1 2 2 > $O/log 2>&1
python - $O/s_kernel_stats.csv $1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
W = int(sys.argv[2]); steps = 3
tot = 0; out = []
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[:44]
    if name.startswith("synth") or "FillFunctor" in name or name.startswith("pack_planes") or name.startswith("scan_u64"): continue
    t = float(r["TotalDurationNs"]) / 1e6 / steps / W
    tot += t; out.append((t, name, int(r["Calls"])))
print("W=%d: GPU kernel time per rank and step = %.2f ms" % (W, tot))
for t, n, c in sorted(out, reverse=True)[:14]: print("   %-46s %8.3f ms  (%d calls)" % (n, t, c))
PY
