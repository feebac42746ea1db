#!/bin/bash
# variants A/B with the finalize stage shown; a few finalize parity tests on the first variant named
b() { cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so; python bench.py --steps 150 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], j['stage_ms_last_step'], j['graph']['nodes'])"; }
for i in 1 2 3; do for v in "$@"; do b $v; done; done
cp scratch/variants/$1.so rust_mdbg_amd/libmdbg_hip.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_c.py tests/test_gpu_dist_scale.py -m gpu -x -q 2>&1 | tail -3
