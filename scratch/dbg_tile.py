import os, sys
sys.path.insert(0, '/root/repo')
import rust_mdbg_amd as R
nreads = 466666
m = R.Mdbg(35, 12, 0.002, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nreads)
for dbg, stag in ((0, 3), (8, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 6), (0, 10)):
    os.environ["MDBG_TILE_DBG"] = str(dbg); os.environ["MDBG_STAGGER"] = str(stag)
    m.reset(0)
    try:
        m.sketch_device(db, do, nreads, nb, 0)
    except Exception as e:
        print("dbg", dbg, "error", e)
    st = m.stats()
    print("dbg", dbg, "stagger", stag, "tile ms %.3f sketch ms %.3f" % (st["ms_sketch_tile"], st["ms_sketch"]), "mins", st["n_minimizers"], "slow", st["n_slow_tiles"], flush=True)
