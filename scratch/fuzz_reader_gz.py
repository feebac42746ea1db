"""extra seeds of tests/test_reader_cpu.py's gzip window test: random records (quality lines starting with '@' / '+' / '>', CRLF, multi-line FASTA, missing last newline),
random container (gzip level / BGZF block size / several members), random thread count and batch size; the records must equal the one-thread reader's on the plain file.
usage: python scratch/fuzz_reader_gz.py [seeds] [first_seed]"""
import gzip, os, random, struct, sys, tempfile, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_reader_cpu import random_records, collect, _bgzf

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + n_seeds):
        rnd = random.Random(seed)
        fastq = rnd.random() < 0.6
        kind = rnd.choice(["", "crlf", "multiline"]) if not fastq else rnd.choice(["", "crlf"])
        n = rnd.choice([1, 2, 30, 300, 800])
        data = random_records(rnd, n, fastq, crlf=kind == "crlf", multiline=kind == "multiline")
        if rnd.random() < 0.3:
            data = data.rstrip(b"\r\n")
        if not fastq and rnd.random() < 0.3:
            data = b"junk\n\n" + data
        strip = (not fastq) and kind == "multiline" and rnd.random() < 0.5
        plain = os.path.join(d, "r.fastq" if fastq else "r.fa")
        open(plain, "wb").write(data)
        ref, _ = collect(plain, strip=strip)
        c = rnd.choice(["gzip", "bgzf", "members", "mixed"])
        if c == "gzip":
            z = gzip.compress(data, rnd.choice([0, 1, 6, 9]))
        elif c == "bgzf":
            z = _bgzf(data, rnd.choice([100, 3000, 65280]))
        elif c == "members":
            cuts = sorted(rnd.randrange(0, len(data) + 1) for _ in range(rnd.randrange(1, 6)))
            z = b"".join(gzip.compress(data[a:b], 6) for a, b in zip([0] + cuts, cuts + [len(data)]))
        else:
            h = rnd.randrange(0, len(data) + 1)
            z = _bgzf(data[:h], 5000) + gzip.compress(data[h:], 6)
        p = plain + ".gz"
        open(p, "wb").write(z)
        for _ in range(4):
            threads = rnd.choice([1, 2, 3, 5, 8, 16])
            mb = rnd.choice([1, 500, 4000, 30000, 200000, 1 << 30])
            got, _ = collect(p, max_bases=mb, strip=strip, threads=threads)
            if got != ref:
                bad += 1
                print("MISMATCH seed", seed, "fastq", fastq, kind, "container", c, "threads", threads, "max_bases", mb, "records", len(ref), len(got))
                break
print("%d seeds, %d mismatches" % (n_seeds, bad))
