#!/bin/bash
# the round's last profile pass: rocprofv3 kernel stats of the default bench command, the multik line, full GPU suite, smoke
R=$(pwd); O=$R/gpurun_out/r5final; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o s -- python $R/bench.py --cpu-seconds 0 > $O/stats_default.json 2> $O/stats_default.err
cd $R
python - $O/stats_default/s_kernel_stats.csv > $O/default_cmd_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-seconds 0   (the driver's command without the CPU leg; durations in microseconds)")
print("%-48s %7s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows[:40]:
    print("%-48s %7s %12.1f %12.1f %12.1f %7s" % (r["Name"].split("(")[0][:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
head -12 $O/default_cmd_kernel_stats.txt
python bench.py --gpus 1 --workload human --multik --steps 3 --warmup 1 --cpu-seconds 0 > $O/multik_n1.json 2> $O/multik_n1.err; python -c "
import json; j=json.load(open('$O/multik_n1.json')); print('multik', j['value'], j['ms_per_step'], j['multik_graph_gbases_per_s'])"
find $O -name "*.csv" -size +20M -delete; find $O -name "*.db" -delete
