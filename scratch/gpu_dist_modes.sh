#!/bin/bash
# one-rank runs of bench.py's multi-GPU code paths (RCCL communicator of one rank): step time per driver / mode
for a in "--force-dist" "--force-dist --dist-impl py" "--force-dist --dist-mode route" "--force-dist --genome-mb 375 --coverage 52 -l 14 --density 0.003" "--force-dist --genome-mb 375 --coverage 52 -l 14 --density 0.003 --dist-mode route" "--genome-mb 375 --coverage 52 -l 14 --density 0.003"; do
  echo "== $a"
  timeout 300 python bench.py $a --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['n_gpus'], 'ms/step %.2f' % j['ms_per_step'], j['graph']['nodes'], j['graph']['windows'], j['config']['parallelism'][:90])
except Exception as e: print('FAILED', e)
"
done
