"""PCIe-inclusive ingest from pageable host memory, 1..4 host threads calling mdbg_ingest_batch concurrently on one context"""
import json, sys, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np
import rust_mdbg_amd as R
nr = 100000
m = R.Mdbg(21, 12, 0.003, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nr)
offs = m.to_host(do, (nr + 1) * 8, np.uint64)
bases = m.to_host(db, int(offs[nr]))
m.close()
out = []
for nthreads, nbatch in ((1, 1), (1, 8), (2, 8), (3, 12), (4, 16)):
    cuts = [nr * i // nbatch for i in range(nbatch + 1)]
    parts = [(np.ascontiguousarray(bases[int(offs[a]):int(offs[b])]), (offs[a:b + 1] - offs[a]).astype(np.uint64), a) for a, b in zip(cuts, cuts[1:])]
    with R.Mdbg(21, 12, 0.003, 2, device=0) as m:
        def work(js):
            for b, o, a in js:
                m.ingest(b, o, a)
        def run():
            th = [threading.Thread(target=work, args=(parts[i::nthreads],)) for i in range(nthreads)]
            [t.start() for t in th]; [t.join() for t in th]
            return m.finalize_device().n
        run(); m.reset(0)
        t = time.perf_counter(); n = run(); m.sync(); dt = time.perf_counter() - t
    out.append(dict(threads=nthreads, batches=nbatch, ms=dt * 1e3, gbases_per_s=len(bases) / dt / 1e9, nodes=int(n)))
print(json.dumps(dict(workload="%.3f Gbases from pageable host memory through mdbg_ingest_batch, k=21 l=12 d=0.003" % (len(bases) / 1e9), rows=out)))
