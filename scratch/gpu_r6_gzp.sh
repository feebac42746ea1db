#!/bin/bash
set -u
timeout 900 python scratch/measure_file_pipeline_gz.py 33333 > /dev/null 2>&1
ls -la /tmp/fpgz/ | head
for r in 1 2; do ./scratch/ubench/gz_prof /tmp/fpgz/reads.fa.gz 5; done
