"""sanity: very long k (LDS staging of the insert kernels) against the oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import synth
from oracle import oracle as O
reads = synth.synth_reads(3, 60000, 40, mean_len=30000, sd_len=3000, min_len=20000, max_len=40000, err_ppm=500)
b, o = O.concat_reads(reads)
for k in (500, 2047, 4096):
    g = O.Graph(k, 8, 0.5, 2); g.ingest(b, o); exp = g.finalize(with_edges=False)
    with R.Mdbg(k, 8, 0.5, 2) as m:
        m.ingest(b, o, 0); got = m.finalize()
    ok = got["n_nodes"] == exp["n_nodes"] and np.array_equal(got["keys"], exp["keys"]) and np.array_equal(got["abundance"], exp["abundance"]) and np.array_equal(got["index"], exp["index"])
    print(k, exp["n_windows"], exp["n_nodes"], "OK" if ok else "MISMATCH")
