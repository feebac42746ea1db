"""host-buffer (PCIe-inclusive) ingest: ASCII through mdbg_ingest_batch vs 2-bit packed through mdbg_pack_reads + mdbg_ingest_batch_packed;
the packer is timed separately (it can run on the reader threads, overlapped with the previous batch)"""
import json, sys, time, os
sys.path.insert(0, '.')
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import emit as E
out = {}
n_reads = 466666
m = R.Mdbg(21, 12, 0.003, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
nr = 200000
offs = m.to_host(do, (nr + 1) * 8, np.uint64)
bases = m.to_host(db, int(offs[nr]))
m.close()
gb = len(bases) / 1e9
cores = os.cpu_count()
m = R.Mdbg(21, 12, 0.003, 2, device=0)
m.ingest(bases, offs, 0); n_ref = int(m.finalize_device().n); m.reset(0)          # warm-up (allocations)
t = time.perf_counter(); m.ingest(bases, offs, 0); n = m.finalize_device().n; m.sync(); dt = time.perf_counter() - t
out["ascii"] = dict(ms=dt * 1e3, gbases_per_s=gb / dt)
for th in (1, 8, 32):
    t = time.perf_counter(); pk = E.pack_reads(bases, offs, threads=th); dtp = time.perf_counter() - t
    out["pack_threads_%d" % th] = dict(ms=dtp * 1e3, gbases_per_s=gb / dtp)
m.reset(0)
m.ingest_packed(pk, 0); assert int(m.finalize_device().n) == n_ref; m.reset(0)
t = time.perf_counter(); m.ingest_packed(pk, 0); n = m.finalize_device().n; m.sync(); dt = time.perf_counter() - t
out["packed_prepacked"] = dict(ms=dt * 1e3, gbases_per_s=gb / dt)
m.reset(0)
t = time.perf_counter(); pk = E.pack_reads(bases, offs, threads=32); m.ingest_packed(pk, 0); n = m.finalize_device().n; m.sync(); dt = time.perf_counter() - t
out["packed_including_pack_32_threads"] = dict(ms=dt * 1e3, gbases_per_s=gb / dt)
out["workload"] = "first %d reads (%.3f Gbases) of the bench reads from pageable host memory, k=21 l=12 d=0.003, %d host cores" % (nr, gb, cores)
print(json.dumps(out))
