#!/bin/bash
cd /root/repo; O=gpurun_out/r5last; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --workload human --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err
python -c "
import json
j=json.load(open('$O/bench_default.json')); print('default', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['stage_ms_last_step'], j['cpu_baseline']['matches_gpu'], j['scale_anchor_n1'].get('value'), j['side_errors'])
j=json.load(open('$O/human_n1.json')); print('human', j['value'], j['ms_per_step'], j['stage_ms_last_step'])"
