#!/bin/bash
# round 6: bench.py's N = 8 branch as a dry run (eight processes on this box's one GPU, the exchange staged through host memory over gloo), against one rank on the same data
set -u
R=$(pwd); O=$R/gpurun_out/r6v; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
S="--workload human --genome-mb 160 --steps 2 --warmup 1 --cpu-seconds 0"
say "one"; timeout 600 python bench.py --gpus 1 $S > $O/one.json 2> $O/one.err; say "rc $?"
say "eight"; timeout 900 python bench.py --gpus 8 --comm host $S > $O/eight.json 2> $O/eight.err; say "rc $?"; tail -5 $O/eight.err | cut -c1-300
python - <<PY
import json
a=json.loads(open("$O/one.json").read().strip().splitlines()[-1]); b=json.loads(open("$O/eight.json").read().strip().splitlines()[-1])
print("nodes", a["graph"]["nodes"], b["graph"]["nodes"], "digest equal", a["graph"]["node_digest"]==b["graph"]["node_digest"], "partitions add up", b["graph"]["partitions_add_up"])
print("value", a["value"], b["value"], b["config"]["parallelism"][:80])
print(json.dumps(b["exchange"])[:1500])
PY
say done
