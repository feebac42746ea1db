"""multi-k sweep on resident sketches (BASELINE configs[4] shape, one GPU shard) and host-buffer (PCIe-inclusive) ingest"""
import json, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import rust_mdbg_amd as R
out = {}
n_reads = 466666
m = R.Mdbg(10, 12, 0.003, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
t = time.perf_counter(); m.ingest_device(db, do, n_reads, nb, 0); n0 = m.finalize_device().n; m.sync(); t_first = time.perf_counter() - t
st = m.stats()
rows = [dict(k=10, ms_total=t_first * 1e3, ms_sketch=st["ms_sketch"], ms_insert=st["ms_insert"], ms_finalize=st["ms_finalize"], nodes=int(n0), windows=st["n_windows"], distinct=st["n_distinct"], note="includes the one-off sketch")]
for k in (15, 20, 25, 30, 35, 40):
    t = time.perf_counter(); m.reset(k); n = m.finalize_device().n; m.sync(); dt = time.perf_counter() - t
    st = m.stats()
    rows.append(dict(k=k, ms_total=dt * 1e3, ms_sketch=st["ms_sketch"], ms_insert=st["ms_insert"], ms_finalize=st["ms_finalize"], nodes=int(n), windows=st["n_windows"], distinct=st["n_distinct"]))
out["multik"] = dict(workload="synthetic 140 Mb @50x (7.0 Gbases), l=12 d=0.003 minabund=2, k = 10..40 step 5 (utils/multik:69-78); sketches stay resident, table cleared and refilled per k", bases=nb, rows=rows)
# PCIe-inclusive: host (pageable) buffers -> mdbg_ingest_batch, 1.5 Gbases
nr = 100000
offs = m.to_host(do, (nr + 1) * 8, np.uint64)
bases = m.to_host(db, int(offs[nr]))
m.close()
m = R.Mdbg(21, 12, 0.003, 2, device=0)
m.ingest(bases, offs, 0); m.finalize_device(); m.reset(0)          # warm-up (allocations)
t = time.perf_counter(); m.ingest(bases, offs, 0); n = m.finalize_device().n; m.sync(); dt = time.perf_counter() - t
out["pcie_inclusive"] = dict(workload="first 100000 reads (%.3f Gbases) from pageable host memory through mdbg_ingest_batch, k=21 l=12 d=0.003" % (len(bases) / 1e9),
                             ms=dt * 1e3, gbases_per_s=len(bases) / dt / 1e9, nodes=int(n))
print(json.dumps(out))
