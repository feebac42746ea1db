#!/bin/bash
# round 4: bench.py's new workloads: dry run of the N>1 path, the human data set through one GPU, the default line with its ASCII leg
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dry_run.py -x -q -m gpu 2>&1 | tail -25 > $O/tests.txt
timeout 600 python bench.py --gpus 1 --workload human --steps 10 --warmup 2 > $O/human_n1.json 2> $O/human_n1.err
timeout 600 python bench.py --gpus 2 --comm host --steps 2 --warmup 1 --genome-mb 750 > $O/human_n2_dry.json 2> $O/human_n2_dry.err
timeout 600 python bench.py > $O/default.json 2> $O/default.err
cat $O/tests.txt; for f in human_n1 human_n2_dry default; do echo "== $f"; cut -c1-1500 $O/$f.json; tail -3 $O/$f.err; done
