"""the .sequences pass (second pass over the input + node lines, LZ4 frames with stored blocks): one writer vs one file per writer thread"""
import sys, time, json, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import pipeline
n = 466666; K, Lm, Dn = 35, 12, 0.002
m = R.Mdbg(K, Lm, Dn, 2, device=0)
db, do, nb = m.synth_reads_device(seed=2, genome_len=140_000_000, n_reads=n)
offs = m.to_host(do, (n + 1) * 8, np.uint64); bases = m.to_host(db, int(offs[n])); m.close()
path = "/tmp/reads.fa"
with open(path, "wb", buffering=1 << 24) as f:
    mv = memoryview(bases)
    for r in range(n):
        f.write(b">r%d\n" % r); f.write(mv[int(offs[r]):int(offs[r + 1])]); f.write(b"\n")
del bases, mv
out = {}
for th in (1, 16):
    for fz in glob.glob("/tmp/outs*.sequences"): os.remove(fz)
    t = time.perf_counter()
    c = pipeline.run_file(path, "/tmp/outs", K, Lm, Dn, 2, write_sequences=True, threads=th)
    dt = time.perf_counter() - t
    sz = sum(os.path.getsize(fz) for fz in glob.glob("/tmp/outs*.sequences"))
    out["%d threads" % th] = dict(total_s=round(dt, 3), sequences_pass_s=round(c["seconds_until"]["sequences"], 3), to_gfa_s=c["seconds_until"]["gfa"], sequences_bytes=sz, nodes=c["n_nodes"],
                                  files=len(glob.glob("/tmp/outs*.sequences")))
for fz in glob.glob("/tmp/outs*.sequences") + [path]: os.remove(fz)
print(json.dumps(dict(workload="BASELINE configs[2] shape, 7.0-Gbase uncompressed FASTA in the page cache -> .gfa + .sequences (one line per node: 35 minimizers + the node's bases)", runs=out)))
