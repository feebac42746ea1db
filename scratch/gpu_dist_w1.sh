#!/bin/bash
# one-rank cost of the multi-GPU C layer (RCCL communicator of one rank) for 1 / 2 / 4 chunks per step, against the plain local step
b() { python bench.py --steps 10 --warmup 3 --cpu-seconds 0 "$@" 2>gpurun_out/q_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'step_ms %.4f' % j['ms_per_step'], j.get('stage_ms_last_step'))"; }
b
b --force-dist --chunks 1
b --force-dist --chunks 2
b --force-dist --chunks 4
