#!/bin/bash
# fin_mark in two passes (claim-map mode) against the one-pass kernel (MDBG_FIN_ONE_PASS=1): suite first, then the human step alternating
cd /root/repo; mkdir -p gpurun_out/r5fin
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5fin/gpu_suite.txt 2>&1; tail -3 gpurun_out/r5fin/gpu_suite.txt
h() { env $1 python bench.py --workload human --steps 5 --warmup 1 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'human step %.3f' % j['ms_per_step'], '%.1f' % j['value'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
b() { env $1 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
(for r in 1 2 3; do h MDBG_FIN_ONE_PASS=1; h X=1; done; for r in 1 2; do b MDBG_FIN_ONE_PASS=1; b X=1; done) > gpurun_out/r5fin/fin_two_pass.txt 2>&1; cat gpurun_out/r5fin/fin_two_pass.txt
