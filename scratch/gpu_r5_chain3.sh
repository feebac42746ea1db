#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5x
python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "insertion" > gpurun_out/r5x/t_insertion.txt 2>&1; tail -15 gpurun_out/r5x/t_insertion.txt
