"""the host reader on gzip / BGZF input on THIS machine's cores: makes a HiFi-shaped FASTQ (level-1 deflate: quick to produce), reads it with 1 / 2 / N threads.
usage: python scratch/host_reader_gz.py [reads] [threads]"""
import os, struct, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rust_mdbg_amd import emit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(1)
parts = []
for i in range(n):
    L = int(rng.normal(15000, 1500))
    parts += [b"@read%d/ccs\n" % i, rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes(), b"\n+\n", rng.choice(np.frombuffer(b"~~~~~~~~~~~~nZF:", np.uint8), size=L).tobytes(), b"\n"]
raw = b"".join(parts)
d = "/tmp/host_reader_gz"; os.makedirs(d, exist_ok=True)
c = zlib.compressobj(1, zlib.DEFLATED, 31); open(d + "/r.fq.gz", "wb").write(c.compress(raw) + c.flush())
out = bytearray()
for i in list(range(0, len(raw), 65280)) + [None]:
    chunk = b"" if i is None else raw[i:i + 65280]
    c = zlib.compressobj(1, zlib.DEFLATED, -15); body = c.compress(chunk) + c.flush()
    out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
open(d + "/r.bgzf.fq.gz", "wb").write(out)
t = time.perf_counter(); zlib.decompress(open(d + "/r.fq.gz", "rb").read(), 31); tz = time.perf_counter() - t
print("%d reads, %.1f MB of text; zlib inflate alone: %.0f MB/s" % (n, len(raw) / 1e6, len(raw) / tz / 1e6))
for path, threads in ((d + "/r.fq.gz", 1), (d + "/r.fq.gz", 2), (d + "/r.fq.gz", 4), (d + "/r.fq.gz", 8), (d + "/r.fq.gz", T), (d + "/r.bgzf.fq.gz", 1), (d + "/r.bgzf.fq.gz", 4), (d + "/r.bgzf.fq.gz", T)):
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); nb = 0
        with emit.Reader(path, threads=threads) as r:
            for bases, offs in r.batches(max_bases=64 << 20, copy=False):
                nb += int(offs[-1])
        best = min(best, time.perf_counter() - t)
    print("%s threads=%d: %.1f Mbases in %.3f s = %.0f Mbases/s = %.0f MB/s of text" % (os.path.basename(path), threads, nb / 1e6, best, nb / 1e6 / best, len(raw) / 1e6 / best))
