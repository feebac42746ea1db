"""host reader on a compressed file: Mbases/s and MB/s of inflated text.  usage: python scratch/bench_reader_gz.py <file> [threads]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rust_mdbg_amd import emit
path = sys.argv[1]; threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for rep in range(2):
    t = time.perf_counter(); nb = nr = 0
    with emit.Reader(path, threads=threads) as r:
        for bases, offs in r.batches(max_bases=64 << 20, copy=False):
            nb += int(offs[-1]); nr += len(offs) - 1
    dt = time.perf_counter() - t
    print("%s threads=%d: %d reads, %.1f Mbases in %.3f s = %.1f Mbases/s" % (os.path.basename(path), threads, nr, nb / 1e6, dt, nb / 1e6 / dt))
