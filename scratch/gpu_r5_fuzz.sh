#!/bin/bash
# fuzz the two-stage insertion: plain, with a two-bit fingerprint (most probes meet another key), without links, on recycled poisoned memory
cd /root/repo; mkdir -p gpurun_out/r5fz
(echo "== plain"; timeout 300 python scratch/fuzz_nodes_edges.py 1000 200 2>&1 | tail -1
 echo "== MDBG_WEAK_FP"; MDBG_WEAK_FP=1 timeout 300 python scratch/fuzz_nodes_edges.py 2000 200 2>&1 | tail -1
 echo "== MDBG_WEAK_FP MDBG_POISON"; MDBG_WEAK_FP=1 MDBG_POISON=1 timeout 300 python scratch/fuzz_nodes_edges.py 3000 120 2>&1 | tail -1
 echo "== MDBG_WEAK_FP MDBG_NO_CHAIN"; MDBG_WEAK_FP=1 MDBG_NO_CHAIN=1 timeout 300 python scratch/fuzz_nodes_edges.py 4000 120 2>&1 | tail -1
 echo "== dist MDBG_WEAK_FP"; MDBG_WEAK_FP=1 timeout 300 python scratch/fuzz_dist.py 40 2>&1 | tail -1
 echo "== pipeline MDBG_WEAK_FP"; MDBG_WEAK_FP=1 timeout 300 python scratch/fuzz_pipeline.py 30 2>&1 | tail -1
 echo "== pytest fuzz + dist + table under MDBG_WEAK_FP"; MDBG_WEAK_FP=1 python -m pytest tests/ -x -q -m gpu -k "fuzz or dist or table or parity or multik or config34" 2>&1 | tail -2) > gpurun_out/r5fz/fuzz.txt 2>&1
cat gpurun_out/r5fz/fuzz.txt
