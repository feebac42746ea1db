#!/bin/bash
# chained confirmation in insert_windows_kernel: parity first, then A/B against MDBG_NO_CHAIN=1 (same build)
cd /root/repo; mkdir -p gpurun_out/r5x
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5x/suite_chain.txt 2>&1; tail -4 gpurun_out/r5x/suite_chain.txt
b() { env $1 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
h() { env $1 python bench.py --workload human --steps 3 --warmup 1 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'human step %.3f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
(b MDBG_NO_CHAIN=1; b X=1; b MDBG_NO_CHAIN=1; b X=1; h MDBG_NO_CHAIN=1; h X=1; h MDBG_NO_CHAIN=1; h X=1) > gpurun_out/r5x/chain_ab.txt 2>&1; cat gpurun_out/r5x/chain_ab.txt
