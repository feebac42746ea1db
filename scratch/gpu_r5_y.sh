#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5y
python bench.py --gpus 1 --workload human --cpu-seconds 0 > gpurun_out/r5y/human_n1.json 2> gpurun_out/r5y/human_n1.err
python bench.py --gpus 1 --workload human --force-dist --cpu-seconds 0 > gpurun_out/r5y/human_dist_w1.json 2> gpurun_out/r5y/human_dist_w1.err
MDBG_NO_CHAIN=1 python bench.py --gpus 1 --workload human --force-dist --cpu-seconds 0 > gpurun_out/r5y/human_dist_w1_nochain.json 2>/dev/null
python -c "
import json
for f in ('human_n1','human_dist_w1','human_dist_w1_nochain'):
    j=json.load(open('gpurun_out/r5y/%s.json'%f)); print(f, j['value'], j['ms_per_step'], j['stage_ms_last_step'], (j.get('no_exchange_anchor') or {}).get('ms_per_step'))
"
