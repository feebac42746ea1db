#!/bin/bash
R=$(pwd); O=$R/gpurun_out/fh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python -m pytest $R/tests/test_gpu_config34.py -x -q -m gpu -k whole_genome_streamed > $O/log 2>&1
tail -3 $O/log
python - $O/s_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]: print("%-60s calls %5s total %10.2f ms  max %10.2f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
