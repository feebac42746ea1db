"""Parity at BASELINE.json sizes (GPU).  configs[1] (100k HiFi-shaped reads, k=21 l=12 d=0.003) is compared bit for bit
with the oracle, nodes AND edges; configs[2] (140 Mb @50x, 7 Gbases) is checked through size-independent properties."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FIELDS = ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed")


def test_config2_100k_reads_bit_exact_nodes_and_edges():
    """BASELINE configs[1]: 30 Mb genome, 100,000 reads ~N(15 kb, 1.5 kb), 0.1 % errors, k=21 l=12 d=0.003 minabund=2"""
    import rust_mdbg_amd as R
    k, l, d, a = 21, 12, 0.003, 2
    n_reads = 100000
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=2, genome_len=30_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        got = m.finalize()
        st = m.stats()
        import time
        t0 = time.perf_counter(); ge = m.graph_edges(0.01); t_gpu = time.perf_counter() - t0
        bases = m.to_host(db, nb)
        offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
    assert 1.4e9 < nb < 1.6e9 and st["n_slow_tiles"] == 0
    g = O.Graph(k, l, d, a)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=True)
    assert st["n_minimizers"] == exp["n_minimizers"] and st["n_windows"] == exp["n_windows"]
    assert got["n_nodes"] == exp["n_nodes"] > 100000 and got["n_nodes_before"] == exp["n_nodes_before"]
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), f
    # edge set: the reference's emitter (pure function of the node table) on OUR table == the oracle's end-to-end edges
    edges, removed = O.edges_from_nodes(got)
    exp_edges = sorted(zip(exp["edge_n1"].tolist(), exp["edge_o1"].tolist(), exp["edge_n2"].tolist(), exp["edge_o2"].tolist(), exp["edge_overlap"].tolist()))
    assert edges == exp_edges and len(edges) == exp["n_edges"] > 100000 and removed == exp["presimp_removed"]
    # ... and the product's own host emitter (libmdbg_emit.so) gives the same L-lines
    from rust_mdbg_amd import emit as E
    t0 = time.perf_counter(); pe = E.Emitter().edges(got); t_host = time.perf_counter() - t0
    assert sorted(zip(pe["n1"].tolist(), pe["o1"].tolist(), pe["n2"].tolist(), pe["o2"].tolist(), pe["overlap"].tolist())) == exp_edges
    assert pe["presimp_removed"] == exp["presimp_removed"]
    # ... and so does the GPU edge builder, in the same order
    for f in ("n1", "o1", "n2", "o2", "overlap"):
        assert np.array_equal(ge[f], pe[f]), f
    assert ge["presimp_removed"] == pe["presimp_removed"]
    print("edges: %d nodes -> %d edges; GPU %.1f ms (incl. copy to host), host emitter %.1f ms" % (got["n_nodes"], len(pe["n1"]), t_gpu * 1e3, t_host * 1e3))


def test_config3_size_properties():
    """BASELINE configs[2] size (7 Gbases on the device): invariants that need no oracle run"""
    import rust_mdbg_amd as R
    k, l, d, a = 35, 12, 0.002, 2
    n_reads = 466666
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        one = m.finalize()
        st = m.stats()
        assert st["n_slow_tiles"] == 0 and st["n_tiles"] == -(-nb // st["tile_bases"])
        # 1. node rows are sorted by index, indices are unique and < number of distinct keys
        assert np.all(np.diff(one["index"].astype(np.int64)) > 0) and int(one["index"][-1]) < one["n_nodes_before"]
        # 2. abundance filter and metadata identities (src/main.rs:778: seqlen = last - first + 2; end = last + l)
        assert one["abundance"].min() >= a
        assert np.array_equal(one["seqlen"], (one["src_end"] - one["src_start"] - l + 2).astype(np.uint32))
        assert np.all(one["src_read"] < n_reads)
        # 3. canonical keys: key <= reversed key lexicographically; every minimizer hash <= hash_bound
        kk = one["keys"]
        rev = kk[:, ::-1]
        neq = kk != rev
        first = np.argmax(neq, axis=1)
        rows = np.arange(len(kk))
        assert np.all((kk[rows, first] < rev[rows, first]) | ~neq.any(axis=1))
        assert int(kk.max()) <= O.hash_bound(d)
        # 4. determinism: drop everything and run again -> identical table (atomics / scheduling must not matter)
        m.reset(0)
        m.ingest_device(db, do, n_reads, nb, 0)
        two = m.finalize()
        for f in FIELDS:
            assert np.array_equal(one[f], two[f]), f
        # 4b. the PACKED path (what bench.py times) gives the identical table at full size, and the counts are the ones bench.py checks
        import torch
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()      # the fill ran on torch's stream, the packer runs on the context's
        assert m.pack_device(db, nb, words.data_ptr()) == 0
        m.reset(0)
        m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0)
        three = m.finalize()
        st3 = m.stats()
        for f in FIELDS:
            assert np.array_equal(one[f], three[f]), f
        del words
        import json
        want = [w for w in json.load(open(os.path.join(ROOT, "tests", "golden", "bench_counts.json")))["workloads"] if w.get("reads_per_gpu") == n_reads and w["l"] == l][0]
        assert nb == want["bases_per_gpu"]
        got3 = {"minimizers": st3["n_minimizers"], "windows": st3["n_windows"], "distinct": st3["n_distinct"], "nodes": three["n_nodes"]}
        assert all(got3[f] == want["graph"][f] for f in got3)
        # ... and the recorded node digest (what bench.py compares with the oracle's in every run) is the digest of this very table, computed here in plain numpy
        assert ["0x%016x" % v for v in O.nodes_digest(three["keys"], three["abundance"])] == want["graph"]["node_digest"]
        assert (st["n_minimizers"], st["n_windows"], st["n_distinct"]) == (st3["n_minimizers"], st3["n_windows"], st3["n_distinct"])
        # 5. multi-k on the resident sketches == a fresh context with that k
        m.reset(21)
        k21 = m.finalize()
        st21 = m.stats()
    assert k21["n_nodes"] > one["n_nodes"] and st21["n_minimizers"] == st["n_minimizers"] and st21["n_windows"] > st["n_windows"]


def test_config3_size_routed_equals_local():
    """the multi-GPU code path (route -> exchange -> insert -> cross-rank resolve) on one rank at 7 Gbases equals the local path"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    k, l, d, a = 35, 12, 0.002, 2
    n_reads = 466666
    dev = torch.device("cuda", 0)
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        loc = m.finalize()
        eng = D.GpuEngine(m, torch, dev)
        eng.reset()
        drv = D.DistributedMdbg(eng, D.ThreadComm(D.ThreadWorld(1), 0, torch), torch)
        drv.ingest_device(db, do, n_reads, nb, 0)
        part = drv.finalize()
        tab = D.gather_node_table([{f: (v.cpu().numpy().view(np.uint64) if hasattr(v, "cpu") else v) for f, v in part.items()}])
    assert tab["n_nodes"] == loc["n_nodes"] and tab["n_nodes_before"] == loc["n_nodes_before"]
    assert np.array_equal(tab["keys"], loc["keys"])
    for f in ("index", "abundance", "seqlen", "reversed", "src_read", "src_start", "src_end"):
        assert np.array_equal(tab[f].astype(np.uint64), loc[f].astype(np.uint64)), f
    assert np.array_equal(tab["shift_full"], loc["shift_full"])


def test_batch_larger_than_2_33_bases_equals_split_batches():
    """one batch of > 2^33 bases (64-bit positions everywhere, > 2^18 tiles in one look-back chain) must equal the same reads
    ingested as two batches"""
    import rust_mdbg_amd as R
    k, l, d, a = 35, 12, 0.002, 2
    n_reads = 600000
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=5, genome_len=180_000_000, n_reads=n_reads)
        assert nb > (1 << 33)
        m.ingest_device(db, do, n_reads, nb, 0)
        one = m.finalize()
        st = m.stats()
        assert st["n_tiles"] == -(-nb // st["tile_bases"]) > 262144 and st["n_sketch_tile_launches"] == 1      # (one launch per batch at every BASELINE configuration: the slabs of 277 k tiles take 1.2 of the 6 GB a launch may use)
        offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
        half = n_reads // 2
        cut = int(offs[half]) // 16 * 16                      # the device bases pointer must stay 16-byte aligned
        m.reset(0)
        import torch
        o2 = torch.from_numpy((offs[half:] - cut).astype(np.int64)).cuda()
        torch.cuda.synchronize()
        m.ingest_device(db, do, half, int(offs[half]), 0)
        m.ingest_device(db + cut, o2.data_ptr(), n_reads - half, nb - cut, half)
        two = m.finalize()
    assert one["n_nodes"] == two["n_nodes"] > 400000 and one["n_nodes_before"] == two["n_nodes_before"]
    for f in FIELDS:
        assert np.array_equal(one[f], two[f]), f


def test_config3_size_replicated_chunked_equals_local():
    """replicated-sketch driver with the batch cut into 4 chunks (zero-copy import path, one rank) at 7 Gbases == local"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    k, l, d, a = 35, 12, 0.002, 2
    n_reads = 466666
    dev = torch.device("cuda", 0)
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        loc = m.finalize()
        eng = D.GpuEngine(m, torch, dev)
        eng.reset()                                           # the partition is set on an empty context
        drv = D.ReplicatedMdbg(eng, D.ThreadComm(D.ThreadWorld(1), 0, torch), torch)
        offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
        drv.ingest_device_chunked(db, eng._view(do, (n_reads + 1,)), D.plan_chunks(offs, 4), 0)
        part = drv.finalize()
        tab = D.gather_node_table([{f: (v.cpu().numpy().view(np.uint64) if hasattr(v, "cpu") else v) for f, v in part.items()}])
    assert tab["n_nodes"] == loc["n_nodes"] and tab["n_nodes_before"] == loc["n_nodes_before"]
    assert np.array_equal(tab["keys"], loc["keys"])
    for f in ("index", "abundance", "seqlen", "reversed", "src_read", "src_start", "src_end"):
        assert np.array_equal(tab[f].astype(np.uint64), loc[f].astype(np.uint64)), f


def test_chromosome_scale_reads():
    """reference-genome-like input (--reference use case): a few reads of tens of megabases each, spanning hundreds of tiles"""
    import rust_mdbg_amd as R
    rng = np.random.default_rng(5)
    chrom = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=24_000_000).tobytes()
    reads = [chrom, chrom[5_000_000:17_000_000], b"ACGT" * 10, chrom[::-1][:3_000_000]]
    k, l, d, a = 21, 12, 0.003, 2
    bases, offs = O.concat_reads(reads)
    g = O.Graph(k, l, d, a)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=False)
    with R.Mdbg(k, l, d, a) as m:
        m.ingest(bases, offs, 0)
        got = m.finalize()
        st = m.stats()
    assert exp["n_nodes"] > 50000 and st["n_tiles"] > 500
    assert got["n_nodes"] == exp["n_nodes"] and got["n_nodes_before"] == exp["n_nodes_before"]
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), f
