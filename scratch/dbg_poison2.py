import sys
sys.path.insert(0, '.')
import numpy as np, torch
import rust_mdbg_amd as R
m = R.Mdbg(35, 14, 0.003, 2)
for g in range(3):
    db, do, nb = m.synth_reads_device(seed=1, genome_len=40_000_000, n_reads=17333, first_read=g * 17333)
    m.sync()
    b = m.to_host(db, nb); o = m.to_host(do, 17334 * 8, np.uint64)
    bad = np.nonzero(~np.isin(b, np.frombuffer(b"ACGT", np.uint8)))[0]
    print("shard", g, "nb", nb, "off ok", bool((np.diff(o.astype(np.int64)) > 0).all()), "o[-1]", int(o[-1]), "bad bytes", len(bad), bad[:5], [hex(int(b[i])) for i in bad[:5]], flush=True)
    if len(bad):
        r = int(np.searchsorted(o, bad[0], side="right")) - 1
        print("  first bad in read", r, "at", int(bad[0] - o[r]), "of", int(o[r + 1] - o[r]), "runs:", np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)[0][[0, -1]], flush=True)
