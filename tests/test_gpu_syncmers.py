"""--syncmers (Read::extract_syncmers, src/read.rs:215-352) on the GPU against the oracle and against the committed fixtures of the
independent restatement (tests/golden/independent_syncmer_cases.json)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from test_gpu_parity import rand_reads

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "independent_syncmer_cases.json")))["cases"]


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_syncmers_equal_independent_fixture(ci, packed):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    c = CASES[ci]
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    with R.Mdbg(c["k"], c["l"], c["density"], c["minabund"], reads_already_hpc=c["already_hpc"], syncmer_s=c["syncmer_s"]) as m:
        if packed:
            m.ingest_packed(E.pack_reads(bases, offs), 0)
        else:
            m.ingest(bases, offs, 0)
        sk = m.store_sketch()
        r = m.finalize()
        ge = m.graph_edges(c["presimp"])
    o = sk["off"]
    for i, (pos, hs) in enumerate(c["sketch"]):
        assert sk["pos"][int(o[i]):int(o[i + 1])].tolist() == pos and sk["hashes"][int(o[i]):int(o[i + 1])].tolist() == hs, ("sketch of read", i)
    assert r["n_nodes"] == c["n_nodes"] and r["n_nodes_before"] == c["n_nodes_before"]
    for row, n in enumerate(c["nodes"]):
        assert r["keys"][row].tolist() == n["key"] and int(r["index"][row]) == n["index"] and int(r["abundance"][row]) == n["abundance"]
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and int(r["reversed"][row]) == n["reversed"]
        assert int(r["src_read"][row]) == n["src_read"] and int(r["src_start"][row]) == n["src_start"] and int(r["src_end"][row]) == n["src_end"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in zip(ge["n1"], ge["o1"], ge["n2"], ge["o2"], ge["overlap"]))
    assert got == sorted(c["edges"])


@pytest.mark.parametrize("l,s,d", [(12, 4, 0.05), (14, 6, 0.2), (10, 10, 0.1), (12, 0, 0.01), (31, 16, 1.0), (17, 2, 0.3), (8, 1, 1.0)])
@pytest.mark.parametrize("hpc", [False, True])
def test_gpu_syncmers_random_reads_vs_oracle(l, s, d, hpc):
    """long reads over many tiles and thread segments, homopolymers, N / lower case (resets, no alphabet error), tandem repeats (tied minima)"""
    import rust_mdbg_amd as R
    rng = np.random.default_rng(l * 100 + s)
    reads = rand_reads(700 + l, 25, 0, 90000, hp=0.05)
    reads += [b"", b"ACGT" * 3, b"acgtnACGGTTACAGT" * 40, (b"AC" * 3000) + rand_reads(3, 1, 5000, 5000)[0] + (b"ACGTTGCA" * 2000),
              rand_reads(9, 1, 300000, 300000)[0], b"N" * 50 + rand_reads(4, 1, 2000, 2000)[0]]
    x = bytearray(rand_reads(8, 1, 120000, 120000)[0])
    for p in rng.integers(0, len(x), size=40):
        x[int(p)] = ord("N")
    reads.append(bytes(x))
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, d, hpc, syncmer_s=s)
    assert exp["err"] == 0 and len(exp["hashes"]) > 100
    with R.Mdbg(5, l, d, 2, reads_already_hpc=hpc, syncmer_s=s) as m:
        got = m.sketch(b, o)
    for f in ("off", "hashes", "pos"):
        assert np.array_equal(got[f], exp[f]), f


def test_gpu_syncmer_graph_vs_oracle():
    from rust_mdbg_amd import synth
    import rust_mdbg_amd as R
    reads = synth.synth_reads(13, 300000, 400, mean_len=12000, sd_len=1500, min_len=3000, max_len=20000, err_ppm=2000)
    k, l, s, d, a = 7, 12, 4, 0.05, 2
    b, o = O.concat_reads(reads)
    g = O.Graph(k, l, d, a, syncmer_s=s)
    assert g.ingest(b, o) == 0
    exp = g.finalize(with_edges=False)
    assert exp["n_nodes"] > 500
    with R.Mdbg(k, l, d, a, syncmer_s=s) as m:
        m.ingest(b[:int(o[200])], o[:201], 0)
        m.ingest(b[int(o[200]):], o[200:] - o[200], 200)
        got = m.finalize()
    from test_gpu_parity import assert_nodes_equal
    assert_nodes_equal(got, exp)


@pytest.mark.parametrize("l,s,d", [(5, 5, 1.0), (6, 5, 1.0), (9, 3, 0.5), (12, 4, 1.0), (20, 4, 0.3), (31, 1, 0.2), (31, 13, 1.0), (17, 9, 0.5), (12, 0, 0.3), (16, 8, 1.0)])
@pytest.mark.parametrize("hpc", [False, True])
def test_gpu_syncmer_scan_edges(l, s, d, hpc):
    """the scan formulation of the window-minimum machine (round 6) where its carries travel: thousands of short reads (read starts in every lane's view, first windows, reads
    shorter than l), tandem repeats of every period up to 9 and of lengths on either side of the look-back round (tied minima over hundreds of windows: the fix-up pass, the
    fixed-point loop over a wave, the fallback to the byte-wise machine), one base repeated (homopolymer compression leaves a single base), window sizes 1, 2, 7 .. 31"""
    import random
    import rust_mdbg_amd as R
    rnd = random.Random(1000 * l + s)
    reads = rand_reads(31 * l + s, 1500, 0, 3 * l + 40)                     # short reads back to back
    reads += rand_reads(7 * l + s, 60, 200, 700, hp=0.1)
    for period in range(1, 10):
        unit = bytes(rnd.choice(b"ACGT") for _ in range(period))
        for ln in (40, 300, 1100, 2500, 9000):
            reads.append(rand_reads(period * 100 + ln, 1, 50, 50)[0] + (unit * (ln // period + 1))[:ln] + rand_reads(period * 100 + ln + 1, 1, 80, 80)[0])
    reads.append(b"ACGGT" * 5000 + rand_reads(3, 1, 40000, 40000)[0] + b"TTGCA" * 300 + b"AG" * 700)
    rnd.shuffle(reads)
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, d, hpc, syncmer_s=s)
    assert exp["err"] == 0 and len(exp["hashes"]) > 100
    with R.Mdbg(5, l, d, 2, reads_already_hpc=hpc, syncmer_s=s) as m:
        got = m.sketch(b, o)
    for f in ("off", "hashes", "pos"):
        assert np.array_equal(got[f], exp[f]), f
