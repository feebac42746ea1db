#!/bin/bash
# round 5, file pipeline: why more than 16 - 32 reader threads are slower; ASCII batches at 16 / 32 threads
cd /root/repo; mkdir -p gpurun_out/r5f
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import rust_mdbg_amd as R
n = 466666
m = R.Mdbg(35, 12, 0.002, 2, device=0)
db, do, nb = m.synth_reads_device(seed=2, genome_len=140_000_000, n_reads=n)
offs = m.to_host(do, (n + 1) * 8, np.uint64); bases = m.to_host(db, int(offs[n])); m.close()
with open("/tmp/reads.fa", "wb", buffering=1 << 24) as f:
    mv = memoryview(bases)
    for r in range(n):
        f.write(b">r%d\n" % r); f.write(mv[int(offs[r]):int(offs[r + 1])]); f.write(b"\n")
PY
gcc -O2 -Iinclude examples/mdbg_cli.c -Lrust_mdbg_amd -lmdbg_hip -lmdbg_emit -lpthread -Wl,-rpath,/root/repo/rust_mdbg_amd -o /tmp/mdbg_cli
for t in 16 32 64; do
  for rep in 1 2; do MDBG_READER_TIMING=1 /tmp/mdbg_cli /tmp/reads.fa -k 35 -l 12 --density 0.002 --minabund 2 --prefix /tmp/outc --no-basespace --threads $t --timing 2>&1 | grep -E "window 268|timing:" | tail -4; done
done > gpurun_out/r5f/cli_threads.txt 2>&1
cat gpurun_out/r5f/cli_threads.txt
MR2_PATH=/tmp/reads.fa python scratch/measure_reader2.py 466666 12,16,24,32,48,64 > gpurun_out/r5f/reader2b.json 2> gpurun_out/r5f/reader2b.err; tail -c 1800 gpurun_out/r5f/reader2b.json
python - <<'PY' > gpurun_out/r5f/ascii_pipeline.json
import sys, time, json
sys.path.insert(0, "/root/repo")
from rust_mdbg_amd import pipeline
out = {}
pipeline.run_file("/tmp/reads.fa", "/tmp/out", 35, 12, 0.002, 2, write_sequences=False, threads=16)
for th in (8, 16, 32):
    for packed in (False, True):
        best = None
        for rep in range(3):
            t = time.perf_counter(); c = pipeline.run_file("/tmp/reads.fa", "/tmp/out", 35, 12, 0.002, 2, write_sequences=False, threads=th, packed=packed); dt = time.perf_counter() - t
            if best is None or dt < best[0]: best = (dt, c)
        out["%d threads, %s" % (th, "packed" if packed else "ASCII")] = dict(seconds=round(best[0], 4), gbases_per_s=round(best[1]["n_bases"] / best[0] / 1e9, 2), seconds_until=best[1]["seconds_until"])
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r5f/ascii_pipeline.json
