#!/bin/bash
# round-end evidence: the whole GPU suite, then the profile set of the default bench
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/final_tests.txt | tail -3
bash scratch/gpu_profile_set.sh r3e > gpurun_out/r3e_profile.log 2>&1; tail -15 gpurun_out/r3e_profile.log
