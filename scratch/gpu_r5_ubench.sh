#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5f
g++ -O2 -std=c++17 -pthread -o /tmp/mmap_vs_pread scratch/ubench/mmap_vs_pread.cpp
python - <<'PY'
import numpy as np
rng=np.random.default_rng(1)
with open('/tmp/big.fa','wb') as f:
    for i in range(0,200000,1000):
        arr=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,size=(1000,15000))]
        f.write(b"".join(b">r%d\n"%(i+j)+arr[j].tobytes()+b"\n" for j in range(1000)))
PY
ls -la /tmp/big.fa
/tmp/mmap_vs_pread /tmp/big.fa 1 8 16 32 64 > gpurun_out/r5f/mmap_vs_pread.txt 2>&1
/tmp/mmap_vs_pread /tmp/big.fa 16 32 >> gpurun_out/r5f/mmap_vs_pread.txt 2>&1
cat gpurun_out/r5f/mmap_vs_pread.txt
lscpu | grep -E "Model name|Socket|NUMA|Thread|L2|L3" > gpurun_out/r5f/lscpu.txt; cat gpurun_out/r5f/lscpu.txt; cat /sys/kernel/mm/transparent_hugepage/enabled; uname -r
