import sys, random, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from rust_mdbg_amd import emit as E
def collect(path, max_bases, strip, threads):
    out = []
    with E.Reader(path, strip, threads=threads) as r:
        for b, o in r.batches(max_bases):
            out += [b[int(o[i]):int(o[i + 1])].tobytes() for i in range(len(o) - 1)]
    return out
def collect_packed(path, max_bases, strip, threads):
    out = []
    with E.Reader(path, strip, threads=threads) as r:
        for pk in r.batches_packed(max_bases):
            nb = pk["n_bases"]; w = pk["words"]
            lo = (w & 0xFFFFFFFF).astype(np.uint64); hi = (w >> np.uint64(32)).astype(np.uint64)
            seq = bytearray(nb)
            for q in range(nb):
                c = ((int(lo[q >> 5]) >> (q & 31)) & 1) | (((int(hi[q >> 5]) >> (q & 31)) & 1) << 1)
                seq[q] = b"ACTG"[c]
            for p_, v in zip(pk["exc_pos"], pk["exc_val"]): seq[int(p_)] = int(v)
            o = pk["offsets"]
            out += [bytes(seq[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
    return out
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(300):
    fastq = rnd.random() < 0.4
    parts = []
    if fastq:
        for i in range(rnd.randint(0, 30)):
            ln = rnd.choice([0, 1, 5, 40, 200])
            seq = bytes(rnd.choice(b"ACGTNacgt") for _ in range(ln))
            q = bytes(rnd.choice(b"@+>I!") for _ in range(ln))
            nl = rnd.choice([b"\n", b"\r\n"])
            parts.append(b"@h" + (b"@" if rnd.random() < .3 else b"") + nl + seq + nl + b"+" + nl + q + nl)
            if rnd.random() < 0.1: parts.append(b"\n")
    else:
        if rnd.random() < 0.3: parts.append(rnd.choice([b"junk\n", b"\n\n", b"xx"]))
        for i in range(rnd.randint(0, 30)):
            nl = rnd.choice([b"\n", b"\r\n"])
            parts.append(b">h%d" % i + nl)
            for _ in range(rnd.choice([0, 1, 1, 2, 5])):
                ln = rnd.choice([0, 1, 5, 40, 200])
                parts.append(bytes(rnd.choice(b"ACGTN\rxg") for _ in range(ln)) + nl)
    data = b"".join(parts)
    if rnd.random() < 0.3: data = data.rstrip(b"\n")
    path = "/tmp/fz.fastq" if fastq else "/tmp/fz.fa"
    open(path, "wb").write(data)
    strip = (not fastq) and rnd.random() < 0.5
    ref = collect(path, 1 << 30, strip, 1)
    for threads in (2, 7):
        for mb in (1 << 30, 300, 17):
            got = collect(path, mb, strip, threads)
            gp = collect_packed(path, mb, strip, threads)
            gp1 = collect_packed(path, mb, strip, 1)
            if got != ref or gp != ref or gp1 != ref:
                bad += 1
                print("MISMATCH", it, fastq, strip, threads, mb, len(ref), len(got), len(gp), len(gp1)); open("/tmp/fz_bad_%d" % it, "wb").write(data)
                break
print("done, bad =", bad)
