#!/bin/bash
# round 6: bench line with the syncmer leg, dry-run tests, then the final profile set and the WHOLE GPU suite
set -u
R=$(pwd); O=$R/gpurun_out/r6u; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "dry run tests"; timeout 1500 python -m pytest tests/test_gpu_bench_dry_run.py -x -q > $O/dry.log 2>&1; say "rc $? $(tail -1 $O/dry.log)"
tail -15 $O/dry.log | cut -c1-200
say "bench default"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; say "rc $?"; python -c "
import json;j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'],j['syncmers'],j.get('side_errors'))"
say done
