#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5s
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5s/gpu_suite.txt 2>&1; tail -5 gpurun_out/r5s/gpu_suite.txt
MDBG_POISON=1 python -m pytest tests/ -x -q -m gpu -k "fuzz or dist or table or round5" > gpurun_out/r5s/gpu_suite_poison.txt 2>&1; tail -3 gpurun_out/r5s/gpu_suite_poison.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5s/smoke.txt 2>&1; tail -2 gpurun_out/r5s/smoke.txt
python bench.py > gpurun_out/r5s/bench_default.json 2> gpurun_out/r5s/bench_default.err; python -c "
import json; j=json.load(open('gpurun_out/r5s/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['stage_ms_last_step'], j['cpu_baseline']['matches_gpu'], j['scale_anchor_n1'].get('value'), j['side_errors'])"
