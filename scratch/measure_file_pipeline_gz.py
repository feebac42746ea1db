"""file -> .gfa from COMPRESSED input (the reference's own example is reads-0.00.fa.gz; src/main.rs:163-178 reads gzip through flate2's GzDecoder): the same reads as
an uncompressed FASTA, an ordinary gzip stream (level 6) and a bgzip-style file, through rust_mdbg_amd.pipeline.run_file (reader -> packer -> GPU ingest -> finalize -> GPU
edges -> .gfa), with several reader thread budgets.  Prints one JSON document (profiles/r06_file_pipeline.json).
usage: python scratch/measure_file_pipeline_gz.py [n_reads]"""
import json
import os
import struct
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import pipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33333
K, Lm, Dn = 21, 12, 0.003
m = R.Mdbg(K, Lm, Dn, 2, device=0)
db, do, nb = m.synth_reads_device(seed=2, genome_len=10_000_000, n_reads=n)
offs = m.to_host(do, (n + 1) * 8, np.uint64)
bases = m.to_host(db, int(offs[n]))
m.close()
d = "/tmp/fpgz"
os.makedirs(d, exist_ok=True)
mv = memoryview(bases)
text = b"".join(b">r%d\n" % r + bytes(mv[int(offs[r]):int(offs[r + 1])]) + b"\n" for r in range(n))
del bases, mv
open(d + "/reads.fa", "wb").write(text)
t = time.perf_counter()
c = zlib.compressobj(6, zlib.DEFLATED, 31)
open(d + "/reads.fa.gz", "wb").write(c.compress(text) + c.flush())
t_gz = time.perf_counter() - t


def bgzf_block(chunk):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(chunk) + c.flush()
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))


with ThreadPoolExecutor(32) as ex:
    blocks = list(ex.map(bgzf_block, [text[i:i + 65280] for i in range(0, len(text), 65280)] + [b""]))
open(d + "/reads.bgzf.fa.gz", "wb").write(b"".join(blocks))
text_gb = len(text) / 1e9
del text, blocks
rows = {}
pipeline.run_file(d + "/reads.fa", d + "/out", K, Lm, Dn, 2, write_sequences=False, threads=16)          # warm-up (library load, page cache)
for name, budgets in (("reads.fa", (16,)), ("reads.fa.gz", (1, 2, 8, 16)), ("reads.bgzf.fa.gz", (2, 8, 16, 32, 64))):
    for th in budgets:
        best = None
        for rep in range(3):
            t = time.perf_counter()
            c = pipeline.run_file(d + "/" + name, d + "/out", K, Lm, Dn, 2, write_sequences=False, threads=th)
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, c)
        rows["%s, %d threads" % (name, th)] = dict(seconds=round(best[0], 4), gbases_per_s=round(best[1]["n_bases"] / best[0] / 1e9, 3), text_gb_per_s=round(text_gb / best[0], 3),
                                                   file_mb=round(os.path.getsize(d + "/" + name) / 1e6, 1), nodes=best[1]["n_nodes"], edges=best[1]["n_edges"])
print(json.dumps(dict(workload="%d HiFi-shaped reads (%.2f Gbases, FASTA, one line per read) of a 10-Mb genome, k=21 l=12 d=0.003 minabund=2; file -> reader -> packer -> GPU -> nodes + edges -> .gfa; "
                               "files in the page cache; best of three" % (n, nb / 1e9), host_cores=os.cpu_count(), text_gb=round(text_gb, 3), zlib_level6_compress_seconds=round(t_gz, 1),
                      note="thread budget of an ordinary gzip stream: ONE inflate thread (read-ahead) + the parser / packer threads; BGZF: half of the budget inflates blocks, half parses",
                      pipeline=rows), indent=1))
