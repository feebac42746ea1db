"""End to end on the GPU box: reads-0.00.fa.gz -> host reader -> GPU hot path -> emitter -> .gfa / .sequences, compared with
what the oracle's restatement of the reference produces for BASELINE configs[0] (k=7 l=10 d=0.0008 minabund=2)."""
import os

import pytest

from conftest import GOLDEN
from oracle import oracle as O
from test_emit_cpu import oracle_edges, read_lz4_frame

pytestmark = pytest.mark.gpu


def test_example_file_to_gfa(example_reads, tmp_path):
    from rust_mdbg_amd import pipeline
    prefix = str(tmp_path / "example")
    k, l, d, a = 7, 10, 0.0008, 2
    c = pipeline.run_file(os.path.join(GOLDEN, "reads-0.00.fa.gz"), prefix, k, l, d, a, batch_bases=3_000_000)    # 5 batches
    assert (c["n_reads"], c["n_bases"], c["n_minimizers"], c["n_windows"]) == (657, 14744805, 16069, 12127)
    assert (c["n_nodes_before"], c["n_nodes"], c["n_edges"], c["presimp_removed"]) == (104, 104, 206, 0)
    g = O.Graph(k, l, d, a)
    b, o = O.concat_reads(example_reads)
    g.ingest(b, o)
    r = g.finalize(with_edges=True)
    lines = open(prefix + ".gfa").read().split("\n")
    assert lines[0] == "H\tVN:Z:1.0"
    assert [x for x in lines if x.startswith("S")] == ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (r["index"][i], r["seqlen"][i], r["abundance"][i]) for i in range(104)]
    assert sorted(x for x in lines if x.startswith("L")) == sorted("L\t%d\t%s\t%d\t%s\t%dM" % (x, chr(p), y, chr(q), ov) for x, p, y, q, ov in oracle_edges(r))
    body = [x for x in read_lz4_frame(prefix + ".0.sequences").decode().split("\n")[4:] if x]
    assert len(body) == 104
    by_id = {int(line.split("\t")[0]): line for line in body}
    assert sorted(by_id) == [int(x) for x in r["index"]]
    for i in range(104):
        f = by_id[int(r["index"][i])].split("\t")
        seq = example_reads[int(r["src_read"][i])][int(r["src_start"][i]):int(r["src_end"][i])]
        if r["reversed"][i]:
            seq = O.revcomp(seq)
        assert f[0] == str(r["index"][i]) and f[2] == seq.decode() and f[5] == "(%d, %d)" % tuple(int(v) for v in r["shift_full"][i])


def test_multik_from_one_pass(example_reads, tmp_path):
    """k sweep on resident sketches (utils/multik:69-78 shape): every k must equal a from-scratch oracle run with that k"""
    from rust_mdbg_amd import pipeline
    prefix = str(tmp_path / "mk")
    l, d, a = 10, 0.0008, 2
    ks = [7, 5, 12]
    out = pipeline.run_multik(os.path.join(GOLDEN, "reads-0.00.fa.gz"), prefix, ks, l, d, a, batch_bases=5_000_000)
    b, o = O.concat_reads(example_reads)
    for k in ks:
        g = O.Graph(k, l, d, a)
        g.ingest(b, o)
        r = g.finalize(with_edges=True)
        c = out[k]
        assert (c["n_nodes_before"], c["n_nodes"], c["n_edges"], c["presimp_removed"]) == (r["n_nodes_before"], r["n_nodes"], r["n_edges"], r["presimp_removed"])
        lines = open("%s-k%d.gfa" % (prefix, k)).read().split("\n")
        assert [x for x in lines if x.startswith("S")] == ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (r["index"][i], r["seqlen"][i], r["abundance"][i]) for i in range(r["n_nodes"])]
        assert sorted(x for x in lines if x.startswith("L")) == sorted("L\t%d\t%s\t%d\t%s\t%dM" % (x, chr(p), y, chr(q), ov) for x, p, y, q, ov in oracle_edges(r))
    assert out[7]["n_nodes"] == 104 and out[5]["n_nodes"] > out[12]["n_nodes"]


def test_plain_c_program_over_the_two_abis(example_reads, tmp_path):
    """examples/mdbg_cli.c (gcc, no Python, no torch): reads-0.00.fa.gz -> .gfa / .sequences through include/mdbg_hip.h and
    include/mdbg_emit.h exactly as a foreign host would call them; output identical to the Python pipeline's"""
    import subprocess
    from conftest import ROOT
    from rust_mdbg_amd import pipeline
    exe = str(tmp_path / "mdbg_cli")
    lib = os.path.join(ROOT, "rust_mdbg_amd")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mdbg_cli.c"), "-L" + lib, "-lmdbg_hip", "-lmdbg_emit",
                    "-lpthread", "-Wl,-rpath," + lib, "-o", exe], check=True)
    src = os.path.join(GOLDEN, "reads-0.00.fa.gz")
    r = subprocess.run([exe, src, "-k", "7", "-l", "10", "--density", "0.0008", "--minabund", "2", "--prefix", str(tmp_path / "c")],
                       check=True, capture_output=True, text=True)
    assert "Number of nodes after abundance filter: 104" in r.stdout and "Number of mdBG edges: 206" in r.stdout
    pipeline.run_file(src, str(tmp_path / "py"), 7, 10, 0.0008, 2)
    assert open(str(tmp_path / "c.gfa")).read() == open(str(tmp_path / "py.gfa")).read()
    assert read_lz4_frame(str(tmp_path / "c.0.sequences")) == read_lz4_frame(str(tmp_path / "py.0.sequences"))
    bad = subprocess.run([exe, src, "-k", "1"], capture_output=True, text=True)            # errors are codes + text, not aborts
    assert bad.returncode == 1 and "invalid parameter" in bad.stderr
    # --threads: the uncompressed file is mapped and parsed in pieces, a reader thread runs one batch ahead, batches go in packed
    import gzip
    plain = str(tmp_path / "reads.fa")
    with gzip.open(src) as f, open(plain, "wb") as g:
        g.write(f.read())
    r = subprocess.run([exe, plain, "-k", "7", "-l", "10", "--density", "0.0008", "--minabund", "2", "--prefix", str(tmp_path / "ct"), "--threads", "4", "--timing"],
                       check=True, capture_output=True, text=True)
    assert "Number of nodes after abundance filter: 104" in r.stdout and "timing:" in r.stderr
    assert open(str(tmp_path / "ct.gfa")).read() == open(str(tmp_path / "py.gfa")).read()
    import glob
    body = lambda raw: sorted(x for x in raw.decode().split("\n") if x and not x.startswith("#"))
    parts = sorted(glob.glob(str(tmp_path / "ct.*.sequences")))
    assert len(parts) == 4                                           # one file per writer thread, like the reference's worker threads
    assert sorted(sum((body(read_lz4_frame(pth)) for pth in parts), [])) == body(read_lz4_frame(str(tmp_path / "py.0.sequences")))
    # --lmer-counts: same counters as the Python pipeline with the same counts file
    import collections
    cnt = collections.Counter()
    for rd in example_reads[:40]:
        text = O.encode_rle(rd)[0]
        for i in range(len(text) - 9):
            cnt[bytes(text[i:i + 10])] += 1
    cf = str(tmp_path / "counts.txt")
    with open(cf, "w") as f:
        for w, c in cnt.items():
            if b"N" not in w:
                f.write("%s\t%d\n" % (w.decode(), c))
    pc = pipeline.run_file(plain, str(tmp_path / "pyl"), 7, 10, 0.004, 2, lmer_counts=cf, lmer_counts_min=1, lmer_counts_max=60)
    subprocess.run([exe, plain, "-k", "7", "-l", "10", "--density", "0.004", "--minabund", "2", "--prefix", str(tmp_path / "cl"), "--threads", "3",
                    "--lmer-counts", cf, "--lmer_counts_min", "1", "--lmer_counts_max", "60"], check=True, capture_output=True, text=True)
    assert pc["n_nodes"] > 0 and open(str(tmp_path / "cl.gfa")).read() == open(str(tmp_path / "pyl.gfa")).read()
    # --syncmers -s / --skiphpc: the reference's flags for the other selection scheme (src/main.rs:490-495)
    r = subprocess.run([exe, plain, "-k", "5", "-l", "12", "--density", "0.05", "--minabund", "2", "--prefix", str(tmp_path / "cs"), "--syncmers", "-s", "4",
                        "--skiphpc", "--no-basespace"], check=True, capture_output=True, text=True)
    b, o = O.concat_reads(example_reads)
    g = O.Graph(5, 12, 0.05, 2, already_hpc=True, syncmer_s=4)
    g.ingest(b, o)
    exp = g.finalize(with_edges=True)
    assert exp["n_nodes"] > 100
    assert ("Number of nodes after abundance filter: %d" % exp["n_nodes"]) in r.stdout and ("Number of mdBG edges: %d" % exp["n_edges"]) in r.stdout


@pytest.mark.parametrize("threads", [1, 4])
def test_reader_thread_is_released_when_the_consumer_fails(tmp_path, threads):
    """an error in the GPU stage (here: a byte outside ACGTN) must not leave the reader (and, with threads > 1, the packer) thread blocked"""
    import threading
    import time
    import rust_mdbg_amd as R
    from rust_mdbg_amd import pipeline
    p = str(tmp_path / "bad.fa")
    with open(p, "w") as f:
        for i in range(400):
            f.write(">r%d\n%s\n" % (i, ("ACGTTGCA" * 500) if i != 3 else ("ACGTTGCA" * 200 + "X" + "ACGTTGCA" * 200)))
    before = threading.active_count()
    with pytest.raises(R.MdbgError) as ei:
        pipeline.run_file(p, str(tmp_path / "out"), 5, 10, 0.02, 2, batch_bases=200_000, threads=threads)      # many batches: the reader runs ahead
    assert ei.value.code == -2
    deadline = time.time() + 5
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before


def test_multik_with_contig_feedback(tmp_path):
    """utils/multik:69-78 with its feedback loop: round i+1 sees [contigs of round i, twice] + reads, in that order.  The GPU side
    keeps the reads' sketches, forgets the previous contigs (mdbg_rewind) and ingests the new ones below the reads' ordinals; every
    round must equal a from-scratch oracle run over the concatenated input in the reference's file order."""
    from rust_mdbg_amd import pipeline, synth
    reads = synth.synth_reads(5, 400000, 120, mean_len=14000, sd_len=1500, min_len=5000, max_len=20000, err_ppm=1000)
    fa = str(tmp_path / "reads.fa")
    with open(fa, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">r%d\n" % i + s + b"\n")
    l, d, a = 12, 0.003, 2
    ks = [10, 15, 20]
    given = {}

    def contigs_fn(k, gfa_path, nodes):
        # stand-in for magic_simplify: sequences that depend on the round, one of them too short to pass `seqtk seq -L 100000`
        assert os.path.exists(gfa_path) and nodes["n_nodes"] > 0
        j = ks.index(k)
        c = [b"".join(reads[10 * j:10 * j + 9]), reads[j], O.revcomp(b"".join(reads[40 + 8 * j:40 + 8 * j + 8]))]
        given[k] = c
        return c

    out = pipeline.run_multik(fa, str(tmp_path / "mk"), ks, l, d, a, batch_bases=400_000, contigs_fn=contigs_fn)
    prev = []
    for k in ks:
        kept = [c for c in prev if len(c) >= 100000]
        assert out[k]["n_contigs"] == len(kept) and (not prev or len(kept) == 2)
        b, o = O.concat_reads(kept + kept + reads)
        g = O.Graph(k, l, d, a)
        g.ingest(b, o)
        r = g.finalize(with_edges=True)
        c = out[k]
        assert (c["n_nodes_before"], c["n_nodes"], c["n_edges"], c["presimp_removed"]) == (r["n_nodes_before"], r["n_nodes"], r["n_edges"], r["presimp_removed"])
        lines = open("%s-k%d.gfa" % (str(tmp_path / "mk"), k)).read().split("\n")
        assert [x for x in lines if x.startswith("S")] == ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (r["index"][i], r["seqlen"][i], r["abundance"][i]) for i in range(r["n_nodes"])]
        assert sorted(x for x in lines if x.startswith("L")) == sorted("L\t%d\t%s\t%d\t%s\t%dM" % (x, chr(p), y, chr(q), ov) for x, p, y, q, ov in oracle_edges(r))
        prev = given[k]
    assert out[15]["n_nodes"] != out[10]["n_nodes"]


def test_file_pipeline_with_lmer_counts(example_reads, tmp_path):
    """--lmer-counts through the file pipeline: counts file -> selection -> filtered sketch -> graph, against the oracle"""
    import collections
    from rust_mdbg_amd import pipeline
    k, l, d, a = 7, 10, 0.0008 * 8, 2
    cnt = collections.Counter()
    for r in example_reads[:60]:
        text = O.encode_rle(r)[0]
        for i in range(len(text) - l + 1):
            cnt[bytes(text[i:i + l])] += 1
    lines = [(w, c) for w, c in cnt.items() if b"N" not in w]
    p = str(tmp_path / "counts.txt")
    with open(p, "w") as f:
        for w, c in lines:
            f.write("%s %d\n" % (w.decode(), c))
    out = pipeline.run_file(os.path.join(GOLDEN, "reads-0.00.fa.gz"), str(tmp_path / "lc"), k, l, d, a, batch_bases=5_000_000,
                            lmer_counts=p, lmer_counts_min=1, lmer_counts_max=50, write_sequences=False)
    om = O.LmerMap(lines, l, d, 1, 50)
    b, o = O.concat_reads(example_reads)
    g = O.Graph(k, l, d, a, lmer_map=om)
    g.ingest(b, o)
    r = g.finalize(with_edges=True)
    assert (out["n_minimizers"], out["n_nodes_before"], out["n_nodes"], out["n_edges"]) == (r["n_minimizers"], r["n_nodes_before"], r["n_nodes"], r["n_edges"])
    plain = O.sketch(b, o, l, d)
    assert 0 < out["n_minimizers"] < len(plain["hashes"])


def test_file_pipeline_with_host_threads_equals_single_thread(example_reads, tmp_path):
    """threads > 1: mapped file parsed in pieces, batches packed to 2 bits, three overlapped host stages — same graph files"""
    import gzip
    from rust_mdbg_amd import pipeline
    plain = str(tmp_path / "reads.fa")
    with gzip.open(os.path.join(GOLDEN, "reads-0.00.fa.gz")) as f, open(plain, "wb") as g:
        g.write(f.read())
    a = pipeline.run_file(plain, str(tmp_path / "t1"), 7, 10, 0.0008, 2, batch_bases=3_000_000, threads=1)
    b = pipeline.run_file(plain, str(tmp_path / "t8"), 7, 10, 0.0008, 2, batch_bases=3_000_000, threads=8)
    for f in ("n_reads", "n_bases", "n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "n_edges"):
        assert a[f] == b[f], f
    assert a["n_nodes"] == 104
    assert open(str(tmp_path / "t1") + ".gfa").read() == open(str(tmp_path / "t8") + ".gfa").read()
    # threads > 1 writes one .sequences file per writer thread (the reference's layout): together they hold the same node lines
    import glob
    one = [x for x in read_lz4_frame(str(tmp_path / "t1") + ".0.sequences").decode().split("\n") if x and not x.startswith("#")]
    parts = sorted(glob.glob(str(tmp_path / "t8") + ".*.sequences"))
    assert len(parts) == 8
    many = []
    for pth in parts:
        txt = read_lz4_frame(pth).decode().split("\n")
        assert txt[0] == "# k = 7" and txt[1] == "# l = 10"
        many += [x for x in txt if x and not x.startswith("#")]
    assert sorted(many) == sorted(one) and len(one) == 104
