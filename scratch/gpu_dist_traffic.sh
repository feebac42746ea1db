#!/bin/bash
# bytes per rank and step of the C layer's exchange, segments vs whole sketches (thread ranks on one GPU)
for a in "$@"; do timeout 600 python scratch/measure_dist_traffic.py $a 2>gpurun_out/dt_err.txt | tail -1 | tee -a gpurun_out/dist_traffic.jsonl; tail -2 gpurun_out/dt_err.txt | grep -i "error\|assert" ; done
