#!/bin/bash
# step / stage times of the default bench for several table load factors (MDBG_TABLE_FACTOR_64THS: slots = keys * x / 64)
for F in 72 80 96 128 192; do
  MDBG_TABLE_FACTOR_64THS=$F timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor $F/64', 'step_ms %.4f' % j['ms_per_step'], j['stage_ms_last_step'], 'cap', j['graph']['table_capacity'])"
done
