#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5s
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5s/gpu_suite.txt 2>&1; tail -5 gpurun_out/r5s/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5s/smoke.txt 2>&1; tail -2 gpurun_out/r5s/smoke.txt
