"""debug: W=2 thread world vs one context fed both shards (node counts), and where the sketch time goes"""
import sys, time, json, threading
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rust_mdbg_amd as R
from rust_mdbg_amd import dist as D
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_reads = 466666; dev = torch.device("cuda", 0)
# one context, all shards
with R.Mdbg(35, 12, 0.002, 2, device=0) as m:
    for r in range(W):
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000 * W, n_reads=n_reads, first_read=r * n_reads)
        t = time.perf_counter(); m.sketch_device(db, do, n_reads, nb, r * n_reads); m.sync(); print("single ctx sketch shard", r, 1e3 * (time.perf_counter() - t), "ms", flush=True)
    m.insert_resident(); st = m.stats(); nd = m.finalize_device()
    print("single ctx nodes", int(nd.n), "minimizers", st["n_minimizers"], flush=True)
tw = D.ThreadWorld(W); res = [None] * W; errs = []
def body(r):
    try:
        with R.Mdbg(35, 12, 0.002, 2, device=0) as m:
            db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000 * W, n_reads=n_reads, first_read=r * n_reads)
            eng = D.GpuEngine(m, torch, dev); drv = D.ReplicatedMdbg(eng, D.ThreadComm(tw, r, torch), torch)
            for s in range(3):
                tw.barrier.wait(); drv.reset()
                s0 = m.stats(); t = time.perf_counter()
                drv.ingest_device(db, do, n_reads, nb, r * n_reads)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                n = drv.finalize_device_count(); s1 = m.stats()
                tw.barrier.wait()
                if r == 0: print("step", s, "ingest ms", 1e3 * (t1 - t), dict((k, s1[k] - s0[k]) for k in ("ms_sketch", "ms_sketch_tile", "ms_insert")), "nodes", n, flush=True)
    except BaseException as e:
        errs.append(e); tw.barrier.abort()
th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in th]; [t.join() for t in th]
if errs: raise errs[0]
