"""--lmer-counts on the GPU (mdbg_set_lmer_filter: the density sketch restricted to the l-mers selected from a counts file): the
independent restatement's fixtures through the counts-file reader and the device filter, and seeded larger cases against the oracle —
ASCII and packed input, HPC on and off, N in the reads, l = 32 with the all-G l-mer, an empty selection."""
import collections
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "independent_lmer_cases.json")))["cases"]


def write_counts(path, lines):
    with open(path, "w") as f:
        for w, c in lines:
            f.write("%s\t%d\n" % (w, c))


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_lmer_counts_equal_independent_fixture(ci, packed, tmp_path):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    from oracle import oracle as O
    c = CASES[ci]
    p = str(tmp_path / "counts.txt")
    write_counts(p, c["lmer_lines"])
    codes, _ = E.lmer_filter_from_counts(p, c["l"], c["density"], c["lmer_min"], c["lmer_max"])
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    with R.Mdbg(c["k"], c["l"], c["density"], c["minabund"], reads_already_hpc=c["already_hpc"]) as m:
        m.set_lmer_filter(codes)
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
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and int(r["src_start"][row]) == n["src_start"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in zip(ge["n1"], ge["o1"], ge["n2"], ge["o2"], ge["overlap"]))
    assert got == sorted(c["edges"]) and ge["presimp_removed"] == c["presimp_removed"]


def count_lmers(reads, l, already_hpc):
    from oracle import oracle as O
    cnt = collections.Counter()
    for r in reads:
        text = r if already_hpc else O.encode_rle(r)[0]
        for i in range(len(text) - l + 1):
            w = text[i:i + l]
            if b"N" not in w:
                cnt[bytes(w)] += 1
    return cnt


@pytest.mark.parametrize("seed", range(6))
def test_gpu_lmer_counts_against_oracle_larger(seed, tmp_path):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E, synth
    from oracle import oracle as O
    rnd = random.Random(seed)
    l = [12, 10, 14, 31, 8, 12][seed]
    d = [0.01, 0.05, 0.02, 0.02, 0.1, 0.005][seed]
    k = [10, 8, 5, 6, 12, 7][seed]
    already_hpc = seed == 4
    reads = synth.synth_reads(100 + seed, 150000, 160, mean_len=9000, sd_len=2500, min_len=500, max_len=16000, err_ppm=2000)
    if seed in (1, 3):                                        # N inside reads: packed input takes the exception list, the l-mer is in no counts file
        reads = [r[:len(r) // 2] + b"N" + r[len(r) // 2 + 1:] if i % 7 == 0 else r for i, r in enumerate(reads)]
    cnt = count_lmers(reads, l, already_hpc)
    lines = [(w.decode(), c) for w, c in cnt.items() if rnd.random() > 0.1]        # the counter missed a tenth of the l-mers
    cmin, cmax = [(1, 100000), (2, 100000), (0, 40), (1, 100000), (2, 60), (0, 100000)][seed]
    p = str(tmp_path / "counts.txt")
    write_counts(p, lines)
    codes, ignored = E.lmer_filter_from_counts(p, l, d, cmin, cmax)
    assert ignored == 0
    om = O.LmerMap([(w.encode(), c) for w, c in lines], l, d, cmin, cmax)
    assert len(om.selected()) == len(codes)
    bases, offs = O.concat_reads(reads)
    exp_sk = O.sketch(bases, offs, l, d, already_hpc, lmer_map=om)
    plain = O.sketch(bases, offs, l, d, already_hpc)
    assert 0 < len(exp_sk["hashes"]) < len(plain["hashes"])                        # the filter does something
    g = O.Graph(k, l, d, 2, already_hpc, 0.01, lmer_map=om)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize()
    for packed in (False, True):
        with R.Mdbg(k, l, d, 2, reads_already_hpc=already_hpc) as m:
            m.set_lmer_filter(codes)
            half = len(reads) // 2                                                  # two batches
            for lo, hi in ((0, half), (half, len(reads))):
                b2, o2 = O.concat_reads(reads[lo:hi])
                if packed:
                    m.ingest_packed(E.pack_reads(b2, o2), lo)
                else:
                    m.ingest(b2, o2, lo)
            sk = m.store_sketch()
            r = m.finalize()
            st = m.stats()
        assert np.array_equal(sk["hashes"], exp_sk["hashes"]) and np.array_equal(sk["pos"], exp_sk["pos"]) and np.array_equal(sk["off"], exp_sk["off"])
        assert st["n_minimizers"] == len(exp_sk["hashes"])
        assert r["n_nodes"] == exp["n_nodes"] and r["n_nodes_before"] == exp["n_nodes_before"] and r["n_nodes"] > 0
        for f in ("keys", "index", "abundance", "seqlen", "shift", "src_read", "src_start", "src_end", "reversed"):
            assert np.array_equal(np.asarray(r[f]).astype(np.uint64), np.asarray(exp[f]).astype(np.uint64)), f


def test_gpu_lmer_filter_corner_cases():
    import rust_mdbg_amd as R
    from oracle import oracle as O
    l, d = 32, 1.0
    reads = [b"ACGT" * 40 + b"G" * 40 + b"TGCA" * 30, b"GT" * 60]
    bases, offs = O.concat_reads(reads)
    # reads_already_hpc: the all-G 32-mer exists in read 0; its code is all ones (G = 3), the table's "free" marker
    allg = (1 << 64) - 1
    om = O.LmerMap([(b"G" * 32, 5)], l, d, 0, 100)
    exp = O.sketch(bases, offs, l, d, True, lmer_map=om)
    assert len(exp["hashes"]) == 9                                   # 40 - 32 + 1 windows of G only
    with R.Mdbg(3, l, d, 1, reads_already_hpc=True) as m:
        m.set_lmer_filter(np.array([allg, 0], dtype=np.uint64))       # G x 32 and its reverse complement C x 32 (= A0 C1: 0b0101...) -- 0 is A x 32, harmless
        m.ingest(bases, offs, 0)
        sk = m.store_sketch()
        assert np.array_equal(sk["hashes"], exp["hashes"]) and np.array_equal(sk["pos"], exp["pos"])
    with R.Mdbg(3, l, d, 1, reads_already_hpc=True) as m:
        m.set_lmer_filter(np.zeros(0, dtype=np.uint64))              # empty selection: nothing survives
        m.ingest(bases, offs, 0)
        assert m.stats()["n_minimizers"] == 0 and m.finalize()["n_nodes"] == 0
        with pytest.raises(R.MdbgError) as ei:
            m.set_lmer_filter(None)                                  # not while batches are resident
        assert ei.value.code == -6
        m.reset(0)
        m.set_lmer_filter(None)                                      # off again: the plain density sketch
        m.ingest(bases, offs, 0)
        assert m.stats()["n_minimizers"] == len(O.sketch(bases, offs, l, d, True)["hashes"])
    with R.Mdbg(3, 12, 0.1, 1) as m:
        with pytest.raises(R.MdbgError) as ei:
            m.set_lmer_filter(np.array([1 << 24], dtype=np.uint64))  # a code with bits above 2*l
        assert ei.value.code == -1
    with R.Mdbg(3, 40, 0.1, 1) as m:
        with pytest.raises(R.MdbgError) as ei:
            m.set_lmer_filter(np.array([1], dtype=np.uint64))        # l > 32
        assert ei.value.code == -1
    with R.Mdbg(3, 12, 0.1, 1, syncmer_s=4) as m:
        with pytest.raises(R.MdbgError) as ei:
            m.set_lmer_filter(np.array([1], dtype=np.uint64))        # the filter is part of the density scheme only
        assert ei.value.code == -1
