import sys, random
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import oracle as O
from rust_mdbg_amd import emit as E
def decode(code, l): return "".join("ACTG"[(int(code) >> (2 * (l - 1 - i))) & 3] for i in range(l))
rnd = random.Random(5); bad = 0
for it in range(400):
    l = rnd.choice([3, 5, 8, 12, 16, 31, 32]); d = rnd.choice([0.01, 0.1, 0.5, 0.9, 1.0])
    lines = []
    for _ in range(rnd.randint(0, 300)):
        w = "".join(rnd.choice("ACGT") for _ in range(l if rnd.random() < 0.9 else rnd.choice([l + 1, l + 3])))
        if rnd.random() < 0.05: w = w[:l // 2] + "N" + w[l // 2 + 1:]
        lines.append((w, rnd.choice([0, 1, 2, 3, 5, 50, 100000, 4294967295])))
    if lines and rnd.random() < 0.3: lines += [(O.revcomp(x[0].encode()).decode(), rnd.randint(0, 9)) for x in lines[:5]]
    cmin, cmax = rnd.choice([(2, 100000), (0, 4294967295), (1, 3), (5, 5)])
    p = "/tmp/fl.txt"
    with open(p, "w") as f:
        for w, c in lines: f.write("%s%s%d%s\n" % (w, rnd.choice([" ", "\t", "   "]), c, rnd.choice(["", " ", "\r"])))
    codes, ign = E.lmer_filter_from_counts(p, l, d, cmin, cmax)
    ok_lines = [(w.encode(), c) for w, c in lines if len(w) >= l]          # the oracle (like the reference) cannot take shorter l-mers
    m = O.LmerMap(ok_lines, l, d, cmin, cmax)
    exp = sorted(w.decode() for w, _ in m.selected() if set(w.decode()) <= set("ACGT"))
    got = sorted(decode(x, l) for x in codes)
    if got != exp:
        bad += 1; print("MISMATCH", it, l, d, cmin, cmax, len(got), len(exp)); 
print("bad", bad)
