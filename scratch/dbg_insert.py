import os, sys
sys.path.insert(0, '/root/repo')
import rust_mdbg_amd as R
nreads = 466666
m = R.Mdbg(35, 12, 0.002, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nreads)
for dbg in (0, 0, 8, 1, 2, 3, 4, 7):
    os.environ["MDBG_DBG"] = str(dbg)
    m.reset(0)
    m.ingest_device(db, do, nreads, nb, 0)
    st = m.stats()
    print("dbg", dbg, "insert ms %.3f" % st["ms_insert"], "distinct", st["n_distinct"], flush=True)
