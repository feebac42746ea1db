"""debug: GPU sketch vs oracle on small inputs; prints where they differ"""
import sys, numpy as np
sys.path.insert(0, '.')
import rust_mdbg_amd as R
from oracle import oracle as O
rng = np.random.default_rng(1)
def rnd(n): return rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()
def run(reads, l, d, tag, flags=0):
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, d)
    with R.Mdbg(5, l, d, 2, flags=flags) as m:
        got = m.sketch(b, o)
        st = m.stats()
    eh, ep, eo = exp["hashes"], exp["pos"], exp["off"]
    gh, gp, go = got["hashes"], got["pos"], got["off"]
    ok = len(eh) == len(gh) and np.array_equal(eh, gh) and np.array_equal(ep, gp) and np.array_equal(eo, go)
    print(tag, "l=%d d=%g exp=%d got=%d %s" % (l, d, len(eh), len(gh), "OK" if ok else "MISMATCH"), flush=True)
    if not ok:
        es = set(zip(eh.tolist(), ep.tolist())); gs = set(zip(gh.tolist(), gp.tolist()))
        print("   missing %d, extra %d, off equal %s" % (len(es - gs), len(gs - es), np.array_equal(eo, go)))
        n = min(len(eh), len(gh))
        bad = np.nonzero((eh[:n] != gh[:n]) | (ep[:n] != gp[:n]))[0]
        if len(bad): i = bad[0]; print("   first diff at", i, "exp", hex(eh[i]), ep[i], "got", hex(gh[i]), gp[i]); print("   exp pos", ep[max(0,i-2):i+4], "got pos", gp[max(0,i-2):i+4])
        miss = sorted(p for h, p in es - gs)[:10]; ext = sorted(p for h, p in gs - es)[:10]
        print("   missing pos", miss, "extra pos", ext)
    return ok
run([rnd(2000)], 12, 0.01, "one small read")
run([rnd(2000)], 12, 0.01, "generic", flags=1)
run([rnd(20000)], 12, 0.01, "20k")
run([rnd(40000)], 12, 0.01, "40k two tiles")
run([rnd(200000)], 12, 0.003, "200k")
run([rnd(20000), rnd(30000), rnd(500)], 12, 0.01, "3 reads")
run([rnd(20000)], 20, 0.01, "l=20")
run([rnd(20000)], 31, 0.01, "l=31")
run([rnd(20000)], 12, 0.2, "dense")
