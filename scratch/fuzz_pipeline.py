import sys, os, gzip, random, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import oracle as O
from rust_mdbg_amd import pipeline
import test_gpu_fuzz as F
from test_emit_cpu import oracle_edges
bad = []
tmp = tempfile.mkdtemp()
for seed in range(int(sys.argv[1])):
    rnd = random.Random(12000 + seed)
    k, l, d, a = rnd.choice([(3, 8, 0.03, 1), (5, 10, 0.01, 2), (7, 12, 0.008, 2), (4, 6, 0.05, 3)])
    reads = [r for r in F.fuzz_reads(rnd, n_reads=rnd.randint(5, 80), genome_len=rnd.choice([3000, 30000]), mean_len=rnd.choice([300, 3000]),
                                     err=rnd.choice([0.0, 0.01]), p_lower=0.0, p_n=rnd.choice([0.0, 0.2]), p_hp=0.0)]
    reads = [r.replace(b"n", b"N") for r in reads]
    fq = rnd.random() < 0.4; gz = rnd.random() < 0.5
    path = os.path.join(tmp, "r%d.%s%s" % (seed, "fq" if fq else "fa", ".gz" if gz else ""))
    out = bytearray()
    for i, r in enumerate(reads):
        if fq:
            out += b"@r%d some comment\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n"
        else:
            out += b">r%d\n" % i
            if rnd.random() < 0.5 or not r:
                out += r + b"\n"
            else:
                w = rnd.choice([60, 80, 1000])
                for j in range(0, len(r), w):
                    out += r[j:j + w] + (b"\r\n" if rnd.random() < 0.2 else b"\n")
    (gzip.open if gz else open)(path, "wb").write(bytes(out))
    try:
        c = pipeline.run_file(path, os.path.join(tmp, "o%d" % seed), k, l, d, a, batch_bases=rnd.choice([2000, 50000, 10 ** 7]), strip_newlines=True)
        g = O.Graph(k, l, d, a); b, o = O.concat_reads(reads); g.ingest(b, o); r = g.finalize(with_edges=True)
        lines = open(os.path.join(tmp, "o%d.gfa" % seed)).read().split("\n")
        assert (c["n_reads"], c["n_nodes"], c["n_edges"]) == (len(reads), r["n_nodes"], r["n_edges"]), (c, r["n_nodes"], r["n_edges"])
        assert [x for x in lines if x.startswith("S")] == ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (r["index"][i], r["seqlen"][i], r["abundance"][i]) for i in range(r["n_nodes"])]
        assert sorted(x for x in lines if x.startswith("L")) == sorted("L\t%d\t%s\t%d\t%s\t%dM" % (x, chr(p), y, chr(q), ov) for x, p, y, q, ov in oracle_edges(r))
    except BaseException as e:
        bad.append((seed, fq, gz, repr(e)[:120]))
print("bad", bad)
