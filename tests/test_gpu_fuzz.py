"""Seeded randomized parity sweep: random parameters x random read sets, HIP path (through the C ABI) against
the oracle, bit-exact.  Read sets are sampled from a small random genome in both orientations with
substitutions/indels so that k-min-mers repeat across reads (abundance, first-sighting order, canonical
orientation and the A-th sighting rule all get exercised), with lower-case stretches, N runs, homopolymer
runs and empty / shorter-than-l reads mixed in; ingestion is split into random batches.

Reference behaviour covered: src/read.rs:157-211 (encode_rle, extract_density), src/read.rs:238-257
(read_to_kmers), src/kmer_vec.rs:56-70 (normalize), src/main.rs:62-82 and :603-608 (add_kminmer, filter).
"""
import random

import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import _mdbg, assert_nodes_equal, assert_sketch_equal

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def fuzz_reads(rnd, n_reads, genome_len, mean_len, err, p_lower, p_n, p_hp):
    genome = bytearray(rnd.choice(b"ACGT") for _ in range(genome_len))
    if p_hp:
        i = 0
        while i < genome_len:
            if rnd.random() < p_hp:
                run = rnd.randint(2, 60)
                genome[i:i + run] = bytes([genome[i]]) * min(run, genome_len - i)
                i += run
            i += 1
        genome = genome[:genome_len]
    reads = []
    for _ in range(n_reads):
        r = rnd.random()
        if r < 0.03:
            reads.append(b"")
            continue
        ln = rnd.randint(1, 12) if r < 0.08 else max(1, int(rnd.gauss(mean_len, mean_len / 3)))
        ln = min(ln, genome_len)
        st = rnd.randrange(0, genome_len - ln + 1)
        s = bytearray(genome[st:st + ln])
        if err:
            out = bytearray()
            for c in s:
                e = rnd.random()
                if e < err / 3:
                    continue                                  # deletion
                if e < 2 * err / 3:
                    out.append(rnd.choice(b"ACGT"))           # insertion
                elif e < err:
                    c = rnd.choice(b"ACGT")                   # substitution
                out.append(c)
            s = out
        if rnd.random() < 0.5:
            s = bytearray(bytes(s).translate(COMP)[::-1])
        if p_lower and rnd.random() < p_lower and len(s) > 4:
            a = rnd.randrange(len(s)); b = min(len(s), a + rnd.randint(1, 200))
            s[a:b] = bytes(s[a:b]).lower()
        if p_n and rnd.random() < p_n and len(s) > 4:
            a = rnd.randrange(len(s)); b = min(len(s), a + rnd.randint(1, 30))
            s[a:b] = (b"N" if rnd.random() < 0.995 else b"n") * (b - a)      # 'n' is outside the nthash alphabet: error case
        reads.append(bytes(s))
    return reads


def random_case(seed):
    rnd = random.Random(seed)
    l = rnd.choice([2, 3, 5, 8, 10, 11, 12, 13, 14, 14, 15, 17, 24, 31, 32])
    d = rnd.choice([0.0005, 0.002, 0.003, 0.008, 0.02, 0.05, 0.05, 0.2, 0.2, 0.6, 1.0])
    if l <= 5 and d < 0.02:
        d = 0.05                                           # with 4^l <= 1024 l-mers a tiny density selects nothing
    k = rnd.choice([2, 2, 3, 5, 7, 10, 21, 35, 60, 96])
    A = rnd.choice([1, 1, 2, 2, 2, 3, 4, 8])
    hpc = rnd.random() < 0.3
    reads = fuzz_reads(rnd, n_reads=rnd.randint(1, 120), genome_len=rnd.choice([300, 5000, 40000, 150000]),
                       mean_len=rnd.choice([40, 400, 3000, 20000]), err=rnd.choice([0.0, 0.0, 0.01, 0.08]),
                       p_lower=rnd.choice([0.0, 0.0, 0.0, 0.0, 0.0, 0.02]), p_n=rnd.choice([0.0, 0.0, 0.2]), p_hp=rnd.choice([0.0, 0.02]))
    n_cuts = rnd.randint(0, 4)
    cuts = sorted(rnd.randint(0, len(reads)) for _ in range(n_cuts))
    bounds = [0] + cuts + [len(reads)]
    batches = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    if rnd.random() < 0.3:
        rnd.shuffle(batches)                               # batch order must not matter (ordinals carry the order)
    flags = 1 if rnd.random() < 0.15 else 0                # MDBG_FLAG_FORCE_GENERIC
    return dict(k=k, l=l, d=d, A=A, hpc=hpc, reads=reads, batches=batches, flags=flags)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_sketch_and_nodes(seed):
    R = _mdbg()
    c = random_case(1000 + seed)
    bases, offs = O.concat_reads(c["reads"])
    exp_sk = O.sketch(bases, offs, c["l"], c["d"], already_hpc=c["hpc"])
    g = O.Graph(c["k"], c["l"], c["d"], c["A"], already_hpc=c["hpc"])
    rc_exp = g.ingest(bases, offs)
    assert (rc_exp != 0) == (exp_sk["err"] != 0)
    with R.Mdbg(c["k"], c["l"], c["d"], c["A"], reads_already_hpc=c["hpc"], flags=c["flags"]) as m:
        if rc_exp != 0:                                    # a lower-case base inside a hashed read: the reference panics (nthash)
            with pytest.raises(R.MdbgError) as ei:
                m.sketch(bases, offs)
            assert ei.value.code == -2
            with pytest.raises(R.MdbgError):
                m.ingest_reads(c["reads"], 0)
            return
        got_sk = m.sketch(bases, offs)
        assert_sketch_equal(got_sk, exp_sk)
        for (lo, hi) in c["batches"]:
            m.ingest_reads(c["reads"][lo:hi], lo)
        got = m.finalize()
        st = m.stats()
    exp = g.finalize(with_edges=False)
    assert_nodes_equal(got, exp)
    assert st["n_reads"] == len(c["reads"]) and st["n_bases"] == len(bases)
    assert st["n_minimizers"] == int(exp_sk["off"][-1])


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_edges_from_gpu_nodes(seed):
    """Emitter on the GPU node table == oracle edges (src/main.rs:612-677) on random repeat-rich inputs."""
    from rust_mdbg_amd import emit as E
    R = _mdbg()
    rnd = random.Random(77 + seed)
    k, l, d, A = rnd.choice([(3, 8, 0.03, 1), (5, 10, 0.01, 2), (7, 12, 0.008, 2), (4, 6, 0.05, 3)])
    reads = fuzz_reads(rnd, n_reads=150, genome_len=30000, mean_len=4000, err=0.01, p_lower=0.0, p_n=0.0, p_hp=0.0)
    bases, offs = O.concat_reads(reads)
    g = O.Graph(k, l, d, A, presimp=0.01)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=True)
    with R.Mdbg(k, l, d, A) as m:
        m.ingest_reads(reads, 0)
        got = m.finalize()
    assert_nodes_equal(got, exp)
    exp_edges = sorted(zip(exp["edge_n1"].tolist(), exp["edge_o1"].tolist(), exp["edge_n2"].tolist(), exp["edge_o2"].tolist(),
                           exp["edge_overlap"].tolist()))
    pe = E.Emitter().edges(got, presimp=0.01)
    assert sorted(zip(pe["n1"].tolist(), pe["o1"].tolist(), pe["n2"].tolist(), pe["o2"].tolist(), pe["overlap"].tolist())) == exp_edges
    assert len(exp_edges) == exp["n_edges"] and pe["presimp_removed"] == exp["presimp_removed"]


def long_read_case(seed):
    """few long reads (spanning several 65,536-base tiles) with long homopolymers, N runs and low-complexity stretches: tile
    boundaries, the halo, the warm-up replay (homopolymers longer than the halo / than 4096) and the generic-path hand-over"""
    rnd = random.Random(seed)
    l = rnd.choice([5, 8, 10, 12, 12, 14])
    d = rnd.choice([0.002, 0.005, 0.02, 0.1])
    k = rnd.choice([2, 4, 7, 21])
    A = rnd.choice([1, 2, 3])
    reads = []
    for _ in range(rnd.randint(1, 6)):
        s = bytearray()
        target = rnd.choice([70000, 140000, 300000])
        while len(s) < target:
            r = rnd.random()
            if r < 0.55:
                s += bytes(rnd.choice(b"ACGT") for _ in range(rnd.randint(50, 20000)))
            elif r < 0.75:
                s += bytes([rnd.choice(b"ACGT")]) * rnd.choice([3, 40, 130, 300, 5000, 9000])        # homopolymer run
            elif r < 0.85:
                unit = bytes(rnd.choice(b"ACGT") for _ in range(rnd.randint(2, 6)))
                s += unit * rnd.randint(10, 3000)                                                      # short tandem repeat
            elif r < 0.95:
                s += b"N" * rnd.choice([1, 2, 17, 500])
            else:
                s += bytes(rnd.choice(b"ACGT") for _ in range(rnd.randint(1, 30)))
        reads.append(bytes(s))
    rnd.shuffle(reads)
    return dict(k=k, l=l, d=d, A=A, hpc=rnd.random() < 0.25, reads=reads)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_long_reads_across_tiles(seed):
    R = _mdbg()
    c = long_read_case(7000 + seed)
    bases, offs = O.concat_reads(c["reads"])
    exp_sk = O.sketch(bases, offs, c["l"], c["d"], already_hpc=c["hpc"])
    assert exp_sk["err"] == 0
    g = O.Graph(c["k"], c["l"], c["d"], c["A"], already_hpc=c["hpc"])
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=False)
    with R.Mdbg(c["k"], c["l"], c["d"], c["A"], reads_already_hpc=c["hpc"]) as m:
        assert_sketch_equal(m.sketch(bases, offs), exp_sk)
        m.ingest(bases, offs, 0)
        got = m.finalize()
    assert_nodes_equal(got, exp)


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_multik_on_resident_sketches(seed):
    """mdbg_reset(k) sequences on the resident sketches == a fresh oracle run per k (also after a wrapped-abundance k and
    with batches ingested between the resets)"""
    R = _mdbg()
    rnd = random.Random(4000 + seed)
    l = rnd.choice([5, 8, 10, 12])
    d = rnd.choice([0.01, 0.05, 0.2])
    A = rnd.choice([1, 2, 3])
    reads = fuzz_reads(rnd, n_reads=rnd.randint(20, 120), genome_len=rnd.choice([500, 5000, 40000]), mean_len=rnd.choice([400, 3000]),
                       err=rnd.choice([0.0, 0.02]), p_lower=0.0, p_n=0.0, p_hp=rnd.choice([0.0, 0.02]))
    cut = len(reads) // 2
    ks = [rnd.choice([2, 3, 5, 9, 21, 40]) for _ in range(4)]
    with R.Mdbg(ks[0], l, d, A) as m:
        m.ingest_reads(reads[:cut], 0)
        for step, k in enumerate(ks):
            if step:
                m.reset(k)
            if step == 2:
                m.ingest_reads(reads[cut:], cut)        # more reads arrive between two values of k
            got = m.finalize()
            sub = reads if step >= 2 else reads[:cut]
            assert_nodes_equal(got, oracle_graph(sub, k, l, d, A))


from test_gpu_parity import oracle_graph  # noqa: E402
