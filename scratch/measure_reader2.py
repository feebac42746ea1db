"""host reader throughput, round 5: ASCII batches (mdbg_reader_next) and packed batches (mdbg_reader_next_packed) of the parallel reader on an uncompressed FASTA in
the page cache, per thread count; MDBG_READER_NO_FAST=1 in the environment measures the general parser (rounds 2 - 4).  usage: measure_reader2.py [reads] [threads,...]"""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_mdbg_amd import emit as E
from rust_mdbg_amd.api import PackedBatch
n, ln = int(sys.argv[1]) if len(sys.argv) > 1 else 40000, 15000
ths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8]
path = os.environ.get("MR2_PATH") or "/tmp/mr2_reads_%d.fa" % n      # MR2_PATH: an existing FASTA (e.g. the one measure_file_pipeline.py wrote)
rng = np.random.default_rng(1)
if not os.environ.get("MR2_PATH") and (not os.path.exists(path) or os.path.getsize(path) < n * ln):
    with open(path, "wb") as f:
        for i in range(0, n, 1000):
            arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(1000, ln))]
            f.write(b"".join(b">r%d\n" % (i + j) + arr[j].tobytes() + b"\n" for j in range(1000)))
L = E.load_library()
L.mdbg_reader_next_packed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(PackedBatch)]
def run(threads, packed, max_bases=256 << 20):
    r = E.Reader(path, False, threads=threads)
    t = time.perf_counter(); tot = 0; reads = 0
    pb = PackedBatch() if packed else None
    while True:
        if packed:
            assert L.mdbg_reader_next_packed(r.h, max_bases, C.byref(pb)) == 0
            nn = pb.n_reads
            if nn == 0: break
            tot += C.cast(pb.offsets, C.POINTER(C.c_uint64))[nn]
        else:
            b, o, n_ = C.c_void_p(), C.c_void_p(), C.c_uint64()
            assert L.mdbg_reader_next(r.h, max_bases, C.byref(b), C.byref(o), C.byref(n_)) == 0
            nn = n_.value
            if nn == 0: break
            tot += C.cast(o, C.POINTER(C.c_uint64))[nn]
        reads += nn
    dt = time.perf_counter() - t
    r.close()
    return dict(threads=threads, packed=packed, seconds=round(dt, 4), gbases_per_s=round(tot / dt / 1e9, 2), reads=reads)
run(ths[0], False)      # page cache
print(json.dumps(dict(file_gb=os.path.getsize(path) / 1e9, host_cores=os.cpu_count(), no_fast=bool(os.environ.get("MDBG_READER_NO_FAST")),
                      ascii=[run(t, False) for t in ths], packed=[run(t, True) for t in ths])))
