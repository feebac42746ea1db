#!/bin/bash
# rocprofv3 kernel stats of the DEFAULT bench command (python bench.py): the profile the roofline's avg_launch_ms has to agree with
set -u
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/stats.log
cd $R
find $O -name "*.csv" -size +20M -delete
python - "$O" <<'P'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print('# rocprofv3 --kernel-trace --stats -- python bench.py   (default: --steps 20 --warmup 3)')
for r in rows[:24]: print('%-64s calls %5s avg_us %10.1f min_us %10.1f max_us %10.1f  %6s%%' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
P
tail -1 $O/bench_under_rocprof.json | cut -c1-400
