"""Host emitter (libmdbg_emit.so: edges + presimp, GFA, .sequences) against the oracle's restatement of
src/main.rs:1014-1117 and :693-708.  CPU only: the emitter is a pure function of the node table (+ the reads)."""
import json
import os
import random
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from rust_mdbg_amd import emit as E, synth


def oracle_run(reads, k, l, d, a, presimp=0.01):
    g = O.Graph(k, l, d, a, presimp=presimp)
    b, o = O.concat_reads(reads)
    assert g.ingest(b, o) == 0
    return g.finalize(with_edges=True), b, o


def edge_list(e):
    return sorted(zip(e["n1"].tolist(), e["o1"].tolist(), e["n2"].tolist(), e["o2"].tolist(), e["overlap"].tolist()))


def oracle_edges(r):
    return sorted(zip(r["edge_n1"].tolist(), r["edge_o1"].tolist(), r["edge_n2"].tolist(), r["edge_o2"].tolist(), r["edge_overlap"].tolist()))


def read_lz4_frame(path):
    """minimal LZ4 frame reader (stored and compressed blocks are both legal; we only write stored ones)"""
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x04\x22\x4d\x18"
    flg, bd, hc = raw[4], raw[5], raw[6]
    assert flg >> 6 == 1 and not (flg & 0x0C) and not (flg & 1)           # version 01, no content size/checksum, no dict id
    import xxhash
    assert hc == (xxhash.xxh32(raw[4:6], seed=0).intdigest() >> 8) & 0xFF   # header checksum per the LZ4 frame format
    pos, out = 7, b""
    while True:
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        if sz == 0:
            break
        assert sz & 0x80000000, "compressed block: not expected from this writer"
        n = sz & 0x7FFFFFFF
        out += raw[pos:pos + n]
        pos += n
        if flg & 0x10:
            pos += 4
    assert pos == len(raw)
    return out


def test_example_cfg1_edges_and_files(example_reads, tmp_path):
    gold = json.load(open(os.path.join(GOLDEN, "example_cfg1.json")))
    c = gold["config"]
    r, b, o = oracle_run(example_reads, c["k"], c["l"], c["density"], c["minabund"])
    em = E.Emitter()
    e = em.edges(r)
    assert len(e["n1"]) == gold["n_edges"] == 206 and e["presimp_removed"] == gold["presimp_removed"]
    assert edge_list(e) == oracle_edges(r)
    # GFA text: header, one S-line per node in index order, then the L-lines (src/main.rs:1011,1021,1095)
    gfa = str(tmp_path / "ex.gfa")
    em.write_gfa(gfa)
    lines = open(gfa).read().split("\n")
    assert lines[0] == "H\tVN:Z:1.0" and lines[-1] == ""
    s = [x for x in lines if x.startswith("S")]
    assert s == ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (r["index"][i], r["seqlen"][i], r["abundance"][i]) for i in range(r["n_nodes"])]
    ll = sorted(x for x in lines if x.startswith("L"))
    assert ll == sorted("L\t%d\t%s\t%d\t%s\t%dM" % (a, chr(p), bb, chr(q), ov) for a, p, bb, q, ov in oracle_edges(r))
    # .sequences: LZ4 frame; line = index \t [minimizers] \t sequence \t * \t * \t (s0, s1)   (src/main.rs:702)
    seqp = str(tmp_path / "ex.0.sequences")
    em.write_sequences(seqp, r, c["l"], [(b, o, 0)])
    txt = read_lz4_frame(seqp).decode().split("\n")
    assert txt[0] == "# k = 7" and txt[1] == "# l = 10" and txt[3].startswith("# [node name]\t[list of minimizers]")
    body = [x for x in txt[4:] if x]
    assert len(body) == r["n_nodes"]
    for i, line in enumerate(body):
        f = line.split("\t")
        rd, st, en = int(r["src_read"][i]), int(r["src_start"][i]), int(r["src_end"][i])
        seq = example_reads[rd][st:en]
        if r["reversed"][i]:
            seq = O.revcomp(seq)
        assert f == [str(r["index"][i]), "[" + ", ".join(str(int(x)) for x in r["keys"][i]) + "]", seq.decode(), "*", "*",
                     "(%d, %d)" % (r["shift_full"][i][0], r["shift_full"][i][1])]
    # what to_basespace reads back (src/to_basespace.rs:205-214): column 0 id, column 2 sequence, column 5 the two shifts
    f = body[0].split("\t")
    assert [int(x) for x in f[5][1:-1].split(",")] == [int(v) for v in r["shift_full"][0]]


@pytest.mark.parametrize("k,l,d,a,presimp,cov", [(5, 10, 0.004, 2, 0.01, 30), (9, 12, 0.004, 2, 0.5, 40), (3, 8, 0.02, 1, 0.0, 6),
                                                   (2, 12, 0.005, 2, 0.3, 20), (12, 12, 0.006, 3, 0.2, 50)])
def test_edges_synthetic(k, l, d, a, presimp, cov):
    """repeats (duplicated genome segment) + both strands + errors: several candidates per (k-1)-mer, presimp removals"""
    glen = 60000
    reads = synth.synth_reads(k * 7 + int(presimp * 10), glen, glen * cov // 6000, mean_len=6000, sd_len=1500, min_len=1000, max_len=12000, err_ppm=3000)
    rnd = random.Random(k)
    extra = synth.synth_read(99, 0, glen, mean_len=9000, sd_len=1, min_len=9000, max_len=9000, err_ppm=0)
    reads += [extra[:4000] + extra[2000:7000], O.revcomp(extra), extra] * 2          # tandem duplication, palindromic neighbourhoods
    rnd.shuffle(reads)
    r, b, o = oracle_run(reads, k, l, d, a, presimp)
    e = E.Emitter().edges(r, presimp)
    assert e["presimp_removed"] == r["presimp_removed"]
    assert edge_list(e) == oracle_edges(r)
    assert r["n_edges"] > 20
    if presimp >= 0.2:
        assert r["presimp_removed"] > 0


def test_emit_library_exports():
    L = E.load_library()
    for s in E.EXPORTS:
        assert hasattr(L, s)


def test_write_failures_are_io_errors(example_reads, tmp_path):
    """a short write (disk full: /dev/full) or an unopenable path is MDBG_E_IO (-7), never a truncated file reported as OK"""
    if not os.path.exists("/dev/full"):
        pytest.skip("/dev/full not available")
    gold = json.load(open(os.path.join(GOLDEN, "example_cfg1.json")))
    c = gold["config"]
    r, b, o = oracle_run(example_reads, c["k"], c["l"], c["density"], c["minabund"])
    em = E.Emitter()
    em.edges(r)
    with pytest.raises(RuntimeError, match="-7"):
        em.write_gfa("/dev/full")
    with pytest.raises(RuntimeError, match="-7"):
        em.write_gfa(str(tmp_path / "no_such_dir" / "x.gfa"))
    with pytest.raises(RuntimeError, match="-7"):
        em.write_sequences("/dev/full", r, c["l"], [(b, o, 0)])
    with pytest.raises(RuntimeError, match="-7"):
        em.write_sequences(str(tmp_path / "no_such_dir" / "x.sequences"), r, c["l"], [(b, o, 0)])


def test_parallel_sequences_writer_same_lines(example_reads, tmp_path):
    """write_sequences_parallel: T files written by T threads hold together exactly the lines of the single file"""
    gold = json.load(open(os.path.join(GOLDEN, "example_cfg1.json")))
    c = gold["config"]
    r, b, o = oracle_run(example_reads, c["k"], c["l"], c["density"], c["minabund"])
    em = E.Emitter()
    one = str(tmp_path / "one.0.sequences")
    half = len(o) // 2
    cut = int(o[half])
    batches = [(b[:cut], o[:half + 1], 0), (b[cut:], o[half:] - o[half], half)]
    em.write_sequences(one, r, c["l"], batches)
    paths = em.write_sequences_parallel(str(tmp_path / "par"), r, c["l"], batches, 5)
    assert [os.path.basename(p) for p in paths] == ["par.%d.sequences" % t for t in range(5)]
    want = [x for x in read_lz4_frame(one).decode().split("\n") if x and not x.startswith("#")]
    got = []
    for p in paths:
        txt = read_lz4_frame(p).decode().split("\n")
        assert txt[0].startswith("# k = ")
        got += [x for x in txt if x and not x.startswith("#")]
    assert sorted(got) == sorted(want) and len(want) == r["n_nodes"]
