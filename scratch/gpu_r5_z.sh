#!/bin/bash
# the table clear beside the next step's first tile kernel: deferred launch (default) against the immediate one (MDBG_NO_CLEAR_DEFER=1); timelines + step times
cd /root/repo; mkdir -p gpurun_out/r5z
TAG=_defer bash scratch/gpu_r5_k.sh; TAG=_nodefer EXTRA=MDBG_NO_CLEAR_DEFER=1 bash scratch/gpu_r5_k.sh
(for t in defer nodefer; do echo "== $t"; head -1 gpurun_out/r5k_$t/timeline.txt; tail -15 gpurun_out/r5k_$t/timeline.txt; done) > gpurun_out/r5z/clear_overlap_timelines.txt; cat gpurun_out/r5z/clear_overlap_timelines.txt
h() { env $1 python bench.py --workload human --steps 6 --warmup 2 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'human step %.3f' % j['ms_per_step'], '%.1f Gbases/s' % j['value'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
b() { env $1 python bench.py --steps 150 --warmup 5 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], '%.1f Gbases/s' % j['value'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
(for r in 1 2 3; do h MDBG_NO_CLEAR_DEFER=1; h X=1; done; for r in 1 2 3; do b MDBG_NO_CLEAR_DEFER=1; b X=1; done) > gpurun_out/r5z/clear_overlap.txt 2>&1; cat gpurun_out/r5z/clear_overlap.txt
