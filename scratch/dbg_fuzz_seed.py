"""one seed of tests/test_gpu_fuzz.py::test_fuzz_sketch_and_nodes with the first difference printed: python scratch/dbg_fuzz_seed.py <seed> [...]"""
import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_fuzz as F
from oracle import oracle as O
import rust_mdbg_amd as R
for seed in map(int, sys.argv[1:]):
    c = F.random_case(1000 + seed)
    print("seed", seed, {k: v for k, v in c.items() if k not in ("reads",)}, "reads", len(c["reads"]), "bases", sum(len(r) for r in c["reads"]))
    bases, offs = O.concat_reads(c["reads"])
    exp_sk = O.sketch(bases, offs, c["l"], c["d"], already_hpc=c["hpc"])
    with R.Mdbg(c["k"], c["l"], c["d"], c["A"], reads_already_hpc=c["hpc"], flags=c["flags"]) as m:
        try:
            got = m.sketch(bases, offs)
        except Exception as e:
            print("  sketch raised", repr(e), "oracle err", exp_sk["err"]); continue
        print("  oracle err", exp_sk["err"], "n", len(exp_sk["hashes"]), len(got["hashes"]), "stats", {k: v for k, v in m.stats().items() if k in ("n_tiles", "n_slow_tiles", "tile_bases")})
        for f in ("off", "hashes", "pos"):
            a, b = np.asarray(got[f]), np.asarray(exp_sk[f])
            if a.shape != b.shape or not np.array_equal(a, b):
                n = min(len(a), len(b)); d = np.nonzero(a[:n] != b[:n])[0]
                print("  DIFF", f, a.shape, b.shape, "first", (int(d[0]), int(a[d[0]]), int(b[d[0]])) if len(d) else None)
                if f == "off":
                    r = int(d[0]) if len(d) else n
                    print("   read", r - 1, "len", len(c["reads"][r - 1]) if r - 1 < len(c["reads"]) else None, c["reads"][r - 1][:80] if r >= 1 else None)
        try:
            F.test_fuzz_sketch_and_nodes(seed)
            print("  test passes")
        except Exception:
            traceback.print_exc(limit=3)
