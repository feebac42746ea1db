#!/bin/bash
# usage: scratch/gpu_emu_serial_stats.sh <W> [config] : GPU kernel time per rank and step of the emulated W-rank replicate step with the
# ranks' GPU phases serialised (scratch/emulate_world_serial.py), so every kernel's duration is its isolated one. 4 steps (1 warm-up + 3).
R=$(pwd); O=$R/gpurun_out/emus$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/scratch/emulate_world_serial.py $1 1 ${2:-2} > $O/log 2>&1
tail -1 $O/log
python - $O/s_kernel_stats.csv $1 <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
W = int(sys.argv[2]); steps = 4
tot = 0; xch = 0
out = []
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[:44]
    if name.startswith("synth") or "fillBuffer" in name or name.startswith("pack_planes") or name.startswith("scan_u64"): continue      # workload generation
    t = float(r["TotalDurationNs"]) / 1e6 / steps / W
    if "copyBuffer" in name or name.startswith("at::native"): xch += t; continue      # the emulated exchange (device copies) and the harness' tensor ops
    tot += t
    out.append((round(t, 4), name, int(r["Calls"])))
print(json.dumps(dict(world=W, gpu_kernel_ms_per_rank_and_step=round(tot, 3), emulated_exchange_and_harness_ms=round(xch, 3), kernels=sorted(out, reverse=True)[:16])))
PY
