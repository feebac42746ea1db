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


def lz4_block_decode(src):
    """LZ4 block format, written from its specification (test-side decoder, independent of the library's)"""
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1; lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break                                  # the last sequence has literals only
        off = src[i] | src[i + 1] << 8; i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1; ml += b
                if b != 255:
                    break
        ml += 4
        assert 0 < off <= len(out)
        for _ in range(ml):                        # byte by byte: a match may overlap its own output
            out.append(out[-off])
    return bytes(out)


def read_lz4_frame(path, stats=None):
    """minimal LZ4 frame reader: independent blocks, stored or compressed"""
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x04\x22\x4d\x18"
    flg, bd, hc = raw[4], raw[5], raw[6]
    assert flg >> 6 == 1 and (flg & 0x20) and not (flg & 0x0C) and not (flg & 1)   # version 01, independent blocks, no content size/checksum, no dict id
    import xxhash
    assert hc == (xxhash.xxh32(raw[4:6], seed=0).intdigest() >> 8) & 0xFF   # header checksum per the LZ4 frame format
    pos, out = 7, []
    n_comp = n_stored = 0
    while True:
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        if sz == 0:
            break
        n = sz & 0x7FFFFFFF
        assert n <= 4 << 20
        if sz & 0x80000000:
            out.append(raw[pos:pos + n]); n_stored += 1
        else:
            blk = lz4_block_decode(raw[pos:pos + n]); n_comp += 1
            assert len(blk) <= 4 << 20
            # the format's end rules: the last five bytes of a block are literals
            out.append(blk)
        pos += n
        if flg & 0x10:
            pos += 4
    assert pos == len(raw)
    if stats is not None:
        stats.update(compressed_blocks=n_comp, stored_blocks=n_stored, file_bytes=len(raw))
    return b"".join(out)


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


def test_sequences_blocks_are_compressed(tmp_path):
    """the .sequences frame holds COMPRESSED LZ4 blocks (the reference writes its files through lzzzz's frame compressor, src/main.rs:65):
    a test-side decoder reproduces the lines, blocks larger than one 4-MiB block included, and the file is well under the text's size"""
    from rust_mdbg_amd import synth
    k, l, d, a = 12, 12, 0.01, 2
    reads = synth.synth_reads(11, 300000, 1400, mean_len=9000, sd_len=1000, min_len=4000, max_len=15000, err_ppm=1000)
    r, b, o = oracle_run(reads, k, l, d, a)
    assert r["n_nodes"] > 4500
    em = E.Emitter()
    em.edges(r)
    p = str(tmp_path / "big.0.sequences")
    em.write_sequences(p, r, l, [(b, o, 0)])
    st = {}
    lines = read_lz4_frame(p, st).decode().split("\n")
    body = [x for x in lines if x and not x.startswith("#")]
    assert len(body) == r["n_nodes"]
    text_bytes = sum(len(x) + 1 for x in lines)
    assert text_bytes > (4 << 20), "the case must span more than one block"
    assert st["compressed_blocks"] >= 2 and st["stored_blocks"] == 0
    assert st["file_bytes"] < 0.62 * text_bytes, (st, text_bytes)
    # every line parses the way src/to_basespace.rs:203-214 does: index, [minimizers], sequence, *, *, (s0, s1)
    idx = set()
    for x in body[:2000]:
        f = x.split("\t")
        assert len(f) == 6 and f[3] == "*" and f[4] == "*" and f[1].startswith("[") and f[5].startswith("(")
        assert len(f[1][1:-1].split(", ")) == k and set(f[2]) <= set("ACGTN")
        idx.add(int(f[0]))
    assert len(idx) == 2000


def test_large_gfa_written_by_several_threads_is_the_same_text(tmp_path):
    """graphs of 200 k lines and more are formatted by several threads (round 5), each a range of the S lines and of the L lines: the file is the text one
    thread writes (src/main.rs:1011,1021,1095), checked against Python's own formatting of a synthetic table"""
    large_gfa_case(str(tmp_path / "big.gfa"))


def test_large_gfa_in_several_rounds(tmp_path):
    """the writer formats and writes in rounds of a bounded number of lines (round 6: its buffers hold one round, not the file); with rounds of 37,000 lines the test
    table takes five rounds of S lines and six of L lines — the same text (the round size is read once per process: a child)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    child = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nfrom test_emit_cpu import large_gfa_case\nlarge_gfa_case(%r)\nprint('ROUNDS_OK')" % (
        os.path.dirname(here), here, str(tmp_path / "rounds.gfa"))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=dict(os.environ, MDBG_GFA_ROUND_LINES="37000"), timeout=600)
    assert r.returncode == 0 and "ROUNDS_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def large_gfa_case(p):
    import ctypes as C
    from rust_mdbg_amd.api import EdgeList
    rng = np.random.default_rng(5)
    n, m, k = 150_001, 210_007, 3
    nodes = dict(keys=np.zeros((n, k), np.uint64), index=rng.permutation(n).astype(np.uint32), abundance=rng.integers(2, 65535, n).astype(np.uint16),
                 seqlen=rng.integers(1, 4_000_000_000, n, dtype=np.uint64).astype(np.uint32), shift=np.zeros(2 * n, np.uint16), shift_full=np.zeros((n, 2), np.uint64),
                 src_read=np.zeros(n, np.uint64), src_start=np.zeros(n, np.uint64), src_end=np.zeros(n, np.uint64), reversed=np.zeros(n, np.uint8))
    n1 = rng.integers(0, n, m).astype(np.uint32); n2 = rng.integers(0, n, m).astype(np.uint32)
    o1 = rng.choice(np.frombuffer(b"+-", np.uint8), m); o2 = rng.choice(np.frombuffer(b"+-", np.uint8), m)
    ov = rng.integers(0, 100000, m).astype(np.uint32)
    ed = EdgeList(n=m, n1=n1.ctypes.data_as(C.POINTER(C.c_uint32)), o1=o1.ctypes.data_as(C.POINTER(C.c_uint8)), n2=n2.ctypes.data_as(C.POINTER(C.c_uint32)),
                  o2=o2.ctypes.data_as(C.POINTER(C.c_uint8)), overlap=ov.ctypes.data_as(C.POINTER(C.c_uint32)), presimp_removed=0)
    E.Emitter().write_gfa(p, nodes, ed)
    want = ["H\tVN:Z:1.0"] + ["S\t%d\t*\tLN:i:%d\tKC:i:%d" % (nodes["index"][i], nodes["seqlen"][i], nodes["abundance"][i]) for i in range(n)]
    want += ["L\t%d\t%s\t%d\t%s\t%dM" % (n1[i], chr(o1[i]), n2[i], chr(o2[i]), ov[i]) for i in range(m)]
    assert open(p).read() == "\n".join(want) + "\n"


def test_a_gfa_only_node_table_is_refused_where_the_minimizer_lists_are_needed(tmp_path):
    """mdbg_finalize_gfa hands out index / seqlen / abundance and NULL for the rest: write_gfa takes it, the edge builder and the .sequences writer say MDBG_E_PARAM"""
    import ctypes as C
    import numpy as np
    from rust_mdbg_amd import emit as E
    from rust_mdbg_amd.api import Nodes
    L = E.load_library()
    idx, sl, ab = np.array([0, 3], np.uint32), np.array([120, 90], np.uint32), np.array([2, 5], np.uint16)
    nd = Nodes(n=2, k=5, index=idx.ctypes.data_as(C.POINTER(C.c_uint32)), seqlen=sl.ctypes.data_as(C.POINTER(C.c_uint32)), abundance=ab.ctypes.data_as(C.POINTER(C.c_uint16)))
    p = str(tmp_path / "g.gfa")
    assert L.mdbg_emit_write_gfa(p.encode(), C.byref(nd), None) == 0
    assert open(p).read() == "H\tVN:Z:1.0\nS\t0\t*\tLN:i:120\tKC:i:2\nS\t3\t*\tLN:i:90\tKC:i:5\n"
    em = E.Emitter()
    e = E.Edges()
    assert L.mdbg_emit_edges(em.h, C.byref(nd), C.c_float(0.01), C.byref(e)) == -1
    err = C.c_int()
    f = L.mdbg_seqfile_open(str(tmp_path / "x.sequences").encode(), 5, 8, C.byref(err))
    assert f
    b, o = np.frombuffer(b"ACGTACGT", np.uint8), np.array([0, 8], np.uint64)
    assert L.mdbg_seqfile_write_batch(f, C.byref(nd), b.ctypes.data, o.ctypes.data, 1, 0) == -1
    assert L.mdbg_seqfile_close(f) == 0
    nd.index = None
    assert L.mdbg_emit_write_gfa(p.encode(), C.byref(nd), None) == -1
