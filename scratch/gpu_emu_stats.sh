#!/bin/bash
# usage: scratch/gpu_emu_stats.sh <W> : per-kernel GPU time of the emulated W-rank replicate step (all ranks on one GPU), 4 steps
R=$(pwd); O=$R/gpurun_out/emu$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/scratch/emulate_world.py $1 1 > $O/log 2>&1
python - $O/s_kernel_stats.csv $1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
W = int(sys.argv[2]); steps = 4
tot = 0
out = []
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[:44]
    if name.startswith("synth") or "fill" in name.lower(): continue
    t = float(r["TotalDurationNs"]) / 1e6 / steps / W
    tot += t
    out.append((t, name, int(r["Calls"])))
print("W=%d: GPU kernel time per rank and step = %.2f ms" % (W, tot))
for t, n, c in sorted(out, reverse=True)[:12]: print("   %-46s %8.3f ms  (%d calls)" % (n, t, c))
PY
