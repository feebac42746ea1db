#!/bin/bash
# usage: scratch/gpu_stats_bench.sh <name> : rocprofv3 kernel stats of the default bench command + the default bench line itself
set -u
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $O/stats.log 2>&1
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" -size +20M -delete
python - "$O" <<'P'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows[:16]: print('%-60s calls %5s avg_us %10.1f  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
P
cat $O/bench.json
