#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r6smoke; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc $?"; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc $?"; python -c "
import json;j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'],j['roofline']['frac'],j['cpu_baseline']['matches_gpu'],j['scale_anchor_n1']['value'],j['syncmers']['kernel_gbases_per_s'],j.get('side_errors'))"
