#!/bin/bash
# the host reader on gzip input on the GPU box's host: the decoder alone (one thread), the reader with 1 .. 32 threads on an ordinary stream and on BGZF, zlib beside it
set -u
R=$(pwd); O=$R/gpurun_out/r6gz; mkdir -p $O
python - <<'PY'
import numpy as np, zlib, struct, os
rng = np.random.default_rng(1)
parts = []
for i in range(8000):
    L = int(rng.normal(15000, 1500))
    parts += [b"@read%d/ccs\n" % i, rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes(), b"\n+\n", rng.choice(np.frombuffer(b"~~~~~~~~~~~~nZF:", np.uint8), size=L).tobytes(), b"\n"]
raw = b"".join(parts)
d = "/tmp/gzh"; os.makedirs(d, exist_ok=True)
for lvl in (1, 6):
    c = zlib.compressobj(lvl, zlib.DEFLATED, 31); open(d + "/fq_l%d.gz" % lvl, "wb").write(c.compress(raw) + c.flush())
out = bytearray()
for i in list(range(0, len(raw), 65280)) + [None]:
    chunk = b"" if i is None else raw[i:i + 65280]
    c = zlib.compressobj(6, zlib.DEFLATED, -15); body = c.compress(chunk) + c.flush()
    out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
open(d + "/fq_bgzf.gz", "wb").write(out)
open(d + "/fq.fastq", "wb").write(raw)
print("text %.1f MB" % (len(raw) / 1e6), {f: os.path.getsize(d + "/" + f) for f in os.listdir(d)})
PY
for f in fq_l1 fq_l6 fq_bgzf; do ./scratch/ubench/gz_prof /tmp/gzh/$f.gz 5; done
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from rust_mdbg_amd import emit
d = "/tmp/gzh"
size = os.path.getsize(d + "/fq.fastq")
for name in ("fq.fastq", "fq_l1.gz", "fq_l6.gz", "fq_bgzf.gz"):
    for threads in (1, 2, 4, 8, 16, 32):
        for packed in (False, True):
            best = 1e9
            for rep in range(3):
                t = time.perf_counter(); nb = 0
                with emit.Reader(d + "/" + name, threads=threads) as r:
                    if packed:
                        for pk in r.batches_packed(256 << 20): nb += int(pk["n_bases"])
                    else:
                        for bases, offs in r.batches(max_bases=256 << 20, copy=False): nb += int(offs[-1])
                best = min(best, time.perf_counter() - t)
            print("%-12s threads=%2d %s: %.3f s = %.0f Mbases/s = %.0f MB/s of text" % (name, threads, "packed" if packed else "ascii ", best, nb / 1e6 / best, size / 1e6 / best), flush=True)
PY
