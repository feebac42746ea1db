"""end-to-end from a FASTA file on disk (page cache): host reader [-> 2-bit packer] -> GPU ingest -> finalize -> GPU edges -> .gfa,
with 1 / 16 / 64 host threads; and the reader alone (scratch/measure_reader.py)"""
import sys, time, json, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import pipeline
big = len(sys.argv) > 1 and sys.argv[1] == "big"          # BASELINE configs[2] size instead of configs[1]
n = 466666 if big else 100000
K, Lm, Dn = (35, 12, 0.002) if big else (21, 12, 0.003)
m = R.Mdbg(K, Lm, Dn, 2, device=0)
db, do, nb = m.synth_reads_device(seed=2, genome_len=140_000_000 if big else 30_000_000, n_reads=n)
offs = m.to_host(do, (n + 1) * 8, np.uint64); bases = m.to_host(db, int(offs[n])); m.close()
path = "/tmp/reads.fa"
with open(path, "wb", buffering=1 << 24) as f:
    mv = memoryview(bases)
    for r in range(n):
        f.write(b">r%d\n" % r); f.write(mv[int(offs[r]):int(offs[r + 1])]); f.write(b"\n")
del bases, mv
out = {}
pipeline.run_file(path, "/tmp/out", K, Lm, Dn, 2, write_sequences=False)          # warm-up (library load, page cache)
for label, kw in (("1 reader thread, ASCII batches", dict(threads=1)), ("1 thread, packed batches", dict(threads=1, packed=True)),
                  ("16 threads, packed batches", dict(threads=16)), ("32 threads, packed batches", dict(threads=32)), ("64 threads, packed batches", dict(threads=64)),
                  ("16 threads, ASCII batches", dict(threads=16, packed=False)), ("32 threads, ASCII batches", dict(threads=32, packed=False)), ("64 threads, ASCII batches", dict(threads=64, packed=False))):
    best = None
    for rep in range(3):
        t = time.perf_counter()
        c = pipeline.run_file(path, "/tmp/out", K, Lm, Dn, 2, write_sequences=False, **kw)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]: best = (dt, c)
    out[label] = dict(seconds=round(best[0], 4), gbases_per_s=round(best[1]["n_bases"] / best[0] / 1e9, 2), nodes=best[1]["n_nodes"], edges=best[1]["n_edges"], seconds_until=best[1]["seconds_until"])
# the same through the plain-C host (examples/mdbg_cli.c): no Python between the reader, the packer and the ingest call
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/mdbg_cli"
subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "mdbg_cli.c"), "-L" + os.path.join(root, "rust_mdbg_amd"),
                       "-lmdbg_hip", "-lmdbg_emit", "-lpthread", "-Wl,-rpath," + os.path.join(root, "rust_mdbg_amd"), "-o", exe])
cli = {}
for th in (1, 16, 32):
    best = None
    for rep in range(2):
        r = subprocess.run([exe, path, "-k", str(K), "-l", str(Lm), "--density", str(Dn), "--minabund", "2", "--prefix", "/tmp/outc", "--no-basespace", "--threads", str(th), "--timing"],
                           capture_output=True, text=True, check=True)
        line = [x for x in r.stderr.split("\n") if x.startswith("timing:")][0]
        ing = float(line.split("ingest ")[1].split(" s")[0]); tot = float(line.split("to .gfa ")[1].split(" s")[0])
        if best is None or tot < best[1]: best = (ing, tot, line)
    cli["%d threads" % th] = dict(ingest_s=best[0], to_gfa_s=best[1], gbases_per_s=round(nb / best[1] / 1e9, 2), line=best[2])
out_cli = cli
rd = None if big else json.loads(subprocess.check_output([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "measure_reader.py"), "100000", "gz"]).decode().strip().split("\n")[-1])
print(json.dumps(dict(file_gb=os.path.getsize(path) / 1e9, host_cores=os.cpu_count(), workload=("BASELINE configs[2] shape: 466,666 reads, 7.0 Gbases, k=35 l=12 d=0.002" if big else "BASELINE configs[1] shape: 100,000 reads, 1.50 Gbases, k=21 l=12 d=0.003") + " minabund=2, uncompressed FASTA in the page cache; nodes + edges + .gfa",
                      pipeline=out, c_host_mdbg_cli=out_cli, reader_only=rd)))
