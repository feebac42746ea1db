#!/bin/bash
# old insertion kernel (base.so) against the two-stage one (chain.so), with and without links, on one box
cd /root/repo; mkdir -p gpurun_out/r5x
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; env $2 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'step %.4f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
h() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; env $2 python bench.py --workload human --steps 3 --warmup 1 --cpu-seconds 0 --plain 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'human step %.3f' % j['ms_per_step'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'])"; }
(for r in 1 2; do b base X=1; b chain MDBG_NO_CHAIN=1; b chain X=1; done; for r in 1 2; do h base X=1; h chain MDBG_NO_CHAIN=1; h chain X=1; done) > gpurun_out/r5x/chain_ab2.txt 2>&1; cat gpurun_out/r5x/chain_ab2.txt
