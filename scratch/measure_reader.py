"""host reader throughput (no GPU): streaming vs parallel reader on an uncompressed FASTA / FASTQ file in the page cache, and gzip"""
import sys, os, time, json, gzip, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_mdbg_amd import emit as E
n, ln = int(sys.argv[1]) if len(sys.argv) > 1 else 40000, 15000
path = "/tmp/mr_reads.fa"
rng = np.random.default_rng(1)
if not os.path.exists(path) or os.path.getsize(path) < n * ln:
    with open(path, "wb") as f:
        for i in range(0, n, 1000):
            arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(1000, ln))]
            for j in range(1000):
                f.write(b">r%d\n" % (i + j)); f.write(arr[j].tobytes()); f.write(b"\n")
L = E.load_library()
def run(path, threads, max_bases=256 << 20):
    r = E.Reader(path, False, threads=threads)
    t = time.perf_counter(); tot = 0; reads = 0
    while True:
        b, o, nn = C.c_void_p(), C.c_void_p(), C.c_uint64()
        assert L.mdbg_reader_next(r.h, max_bases, C.byref(b), C.byref(o), C.byref(nn)) == 0
        if nn.value == 0: break
        reads += nn.value
        tot += C.cast(o, C.POINTER(C.c_uint64))[nn.value]
    dt = time.perf_counter() - t
    r.close()
    return dict(threads=threads, seconds=round(dt, 3), gbases_per_s=round(tot / dt / 1e9, 2), reads=reads)
def run_pack(path, threads, max_bases=256 << 20):
    t = time.perf_counter(); tot = 0
    with E.Reader(path, False, threads=threads) as r:
        for b, o in r.batches(max_bases, copy=False):
            pk = E.pack_reads(b, o.copy(), threads=threads); tot += len(b)
    dt = time.perf_counter() - t
    return dict(threads=threads, seconds=round(dt, 3), gbases_per_s=round(tot / dt / 1e9, 2))
out = dict(file_gb=os.path.getsize(path) / 1e9, host_cores=os.cpu_count(), fasta=[run(path, t) for t in (1, 1, 2, 4, 8, 16, 32) if t <= max(8, os.cpu_count())])
out["fasta_read_and_pack"] = [run_pack(path, t) for t in (1, 16, 32, 64) if t <= max(8, os.cpu_count())]
if len(sys.argv) > 2:
    gz = path + ".gz"
    if not os.path.exists(gz):
        with open(path, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
            g.write(f.read(300_000_000))
    out["gzip_first_300MB"] = run(gz, 1)
print(json.dumps(out))
