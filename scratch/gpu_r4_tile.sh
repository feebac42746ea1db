#!/bin/bash
# round 4: tile shapes A/B (MDBG_TILE = 4 | 1x1 | 1x4): quick parity tests under each, then the bench alternating
O=gpurun_out/r4b; mkdir -p $O
for T in 1x4 1x1; do
  echo "== parity MDBG_TILE=$T" >> $O/tests.txt
  MDBG_TILE=$T timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 >> $O/tests.txt
done
b() { MDBG_TILE=$1 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 2>$O/err_$1.txt | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step %.4f' % j['ms_per_step'], 'tile(avg) %.4f' % j['roofline']['avg_launch_ms'], {k: round(v, 3) for k, v in j['stage_ms_last_step'].items()}, j['graph']['nodes'], j['graph']['tiles'])"; }
for i in 1 2; do for T in 4 1x4 1x1; do b $T >> $O/ab.txt 2>&1; done; done
cat $O/tests.txt; cat $O/ab.txt; tail -5 $O/err_1x4.txt
