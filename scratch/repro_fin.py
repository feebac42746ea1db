"""finalize of a multi-batch context at the configs[3] shard size: wall time vs the stage's own event time (looking for a host-side stall)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rust_mdbg_amd as R
nb_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_reads = 1300000
m = R.Mdbg(35, 14, 0.003, 2, device=0)
for b in range(nb_batches):
    db, do, nb = m.synth_reads_device(seed=1, genome_len=375_000_000 * nb_batches, n_reads=n_reads, first_read=b * n_reads)
    t = time.perf_counter(); m.ingest_device(db, do, n_reads, nb, b * n_reads); m.sync(); print("batch", b, "ingest %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
for rep in range(2):
    s0 = m.stats(); t = time.perf_counter(); nd = m.finalize_device(); m.sync(); dt = time.perf_counter() - t; s1 = m.stats()
    print("finalize", rep, "wall %.1f ms, stage events %.2f ms, nodes %d, distinct %d" % (dt * 1e3, s1["ms_finalize"] - s0["ms_finalize"], nd.n, s1["n_distinct"]), flush=True)
