#!/bin/bash
# round 5, file pipeline: tail + page-locked batches + one-pass reader
cd /root/repo; mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_round5.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r5f/tests.txt 2>&1; tail -3 gpurun_out/r5f/tests.txt
timeout 900 python scratch/measure_file_pipeline.py big > gpurun_out/r5f/file_pipeline.json 2> gpurun_out/r5f/file_pipeline.err; tail -c 3500 gpurun_out/r5f/file_pipeline.json
gcc -O2 -Iinclude examples/mdbg_cli.c -Lrust_mdbg_amd -lmdbg_hip -lmdbg_emit -lpthread -Wl,-rpath,/root/repo/rust_mdbg_amd -o /tmp/mdbg_cli; MDBG_READER_TIMING=1 /tmp/mdbg_cli /tmp/reads.fa -k 35 -l 12 --density 0.002 --minabund 2 --prefix /tmp/outc --no-basespace --threads 16 --timing 2>&1 | tail -12 > gpurun_out/r5f/cli_timing16.txt; cat gpurun_out/r5f/cli_timing16.txt
MR2_PATH=/tmp/reads.fa python scratch/measure_reader2.py 466666 8,16,24,32,64 > gpurun_out/r5f/reader2.json 2> gpurun_out/r5f/reader2.err; tail -c 1500 gpurun_out/r5f/reader2.json
