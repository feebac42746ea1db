"""end-to-end from a FASTA file on disk: host reader -> mdbg_ingest_batch -> finalize -> GPU edges -> .gfa"""
import sys, time, json, os
sys.path.insert(0, '/root/repo')
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import pipeline
n = 100000
m = R.Mdbg(21, 12, 0.003, 2, device=0)
db, do, nb = m.synth_reads_device(seed=2, genome_len=30_000_000, n_reads=n)
offs = m.to_host(do, (n + 1) * 8, np.uint64); bases = m.to_host(db, int(offs[n])); m.close()
path = "/tmp/reads.fa"
t = time.perf_counter()
with open(path, "wb") as f:
    for r in range(n):
        f.write(b">r%d\n" % r); f.write(bases[int(offs[r]):int(offs[r + 1])].tobytes()); f.write(b"\n")
t_write = time.perf_counter() - t
out = {}
for label, kw in (("nodes+edges+gfa", dict(write_sequences=False)), ("with .sequences (second pass over the file)", dict(write_sequences=True))):
    t = time.perf_counter()
    c = pipeline.run_file(path, "/tmp/out", 21, 12, 0.003, 2, **kw)
    dt = time.perf_counter() - t
    out[label] = dict(seconds=dt, gbases_per_s=c["n_bases"] / dt / 1e9, nodes=c["n_nodes"], edges=c["n_edges"])
print(json.dumps(dict(file_gb=os.path.getsize(path) / 1e9, write_s=t_write, runs=out)))
