"""BASELINE.json configs[3] (synthetic human 3 Gb @52x, k=35 l=14 d=0.003, one GPU's shard of the 8) and configs[4] (the multik
sweep k=10..40, l=12 d=0.003, on the same reads with the sketches kept resident) on one MI355X."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
FIELDS = ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed")
SHARD_READS = 1_300_000          # 3 Gb * 52 / 15 kb / 8 GPUs: ~19.5 Gbases per GPU
SHARD_GENOME = 375_000_000       # an eighth of the genome per GPU (weak scaling)


def test_config4_params_oracle_parity_1gbase():
    """k=35 l=14 d=0.003 minabund=2 on a ~1-Gbase sample of the configs[3] reads (20 Mb of genome at 52x): nodes AND edges, bit for bit"""
    import rust_mdbg_amd as R
    k, l, d, a = 35, 14, 0.003, 2
    n_reads = 69000
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=4, genome_len=20_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        got = m.finalize()
        st = m.stats()
        ge = m.graph_edges(0.01)
        bases = m.to_host(db, nb)
        offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
        # the same reads through the packed path
        import torch
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
        assert m.pack_device(db, nb, words.data_ptr()) == 0
        m.reset(0)
        m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0)
        got_p = m.finalize()
    assert 0.9e9 < nb < 1.2e9 and st["n_slow_tiles"] == 0
    g = O.Graph(k, l, d, a)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=True)
    assert st["n_minimizers"] == exp["n_minimizers"] and st["n_windows"] == exp["n_windows"]
    assert got["n_nodes"] == exp["n_nodes"] > 50000 and got["n_nodes_before"] == exp["n_nodes_before"]
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(got_p[f], exp[f]), "packed " + f
    exp_edges = sorted(zip(exp["edge_n1"].tolist(), exp["edge_o1"].tolist(), exp["edge_n2"].tolist(), exp["edge_o2"].tolist(), exp["edge_overlap"].tolist()))
    assert sorted(zip(ge["n1"].tolist(), ge["o1"].tolist(), ge["n2"].tolist(), ge["o2"].tolist(), ge["overlap"].tolist())) == exp_edges
    assert len(exp_edges) == exp["n_edges"] > 50000 and ge["presimp_removed"] == exp["presimp_removed"]


def _dev(torch, ptr, n, dtype):
    """torch view of n elements of a device buffer owned by the library"""
    import ctypes as C
    from rust_mdbg_amd.dist import _DevArray
    if not isinstance(ptr, int):
        ptr = C.cast(ptr, C.c_void_p).value or 0
    if n == 0:
        return torch.empty(0, dtype=dtype, device="cuda")
    ts = {torch.int64: "<i8", torch.int32: "<i4", torch.int16: "<i2", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DevArray(ptr, (n,), ts), device="cuda")


def _node_tensors(torch, nd, k):
    n = int(nd.n)
    return dict(keys=_dev(torch, nd.keys, n * k, torch.int64).clone(), index=_dev(torch, nd.index, n, torch.int32).clone(),
                abundance=_dev(torch, nd.abundance, n, torch.int16).clone(), seqlen=_dev(torch, nd.seqlen, n, torch.int32).clone(),
                src_read=_dev(torch, nd.src_read, n, torch.int64).clone(), src_start=_dev(torch, nd.src_start, n, torch.int64).clone(),
                src_end=_dev(torch, nd.src_end, n, torch.int64).clone(), reversed=_dev(torch, nd.reversed, n, torch.uint8).clone(),
                shift_full=_dev(torch, nd.shift_full, 2 * n, torch.int64).clone())


def test_config4_full_shard_properties_and_multik_sweep():
    """one GPU's shard of configs[3] at full size (1.3 M reads, ~19.5 Gbases, packed in HBM): size-independent invariants of the
    k=35 l=14 d=0.003 node table, then configs[4]: the sweep k=10..40 with l=12 d=0.003 on the same reads, each k re-windowed from the
    resident sketches (mdbg_reset) and compared with a fresh context"""
    import torch
    import rust_mdbg_amd as R
    a = 2
    with R.Mdbg(35, 14, 0.003, a) as m:
        db, do, nb = m.synth_reads_device(seed=1, genome_len=SHARD_GENOME, n_reads=SHARD_READS)
        assert 19.0e9 < nb < 20.0e9
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
        assert m.pack_device(db, nb, words.data_ptr()) == 0
        offs = _dev(torch, do, SHARD_READS + 1, torch.int64).clone()     # the synthetic buffers belong to this context: keep what outlives it
        m.ingest_packed_device(words.data_ptr(), do, SHARD_READS, nb, 0)
        nd = m.finalize_device()
        st = m.stats()
        one = _node_tensors(torch, nd, 35)
        n = int(nd.n)
        assert st["n_slow_tiles"] == 0 and st["n_tiles"] == -(-nb // st["tile_bases"]) and n > 1_000_000
        # 1. rows sorted by index, unique, below the number of distinct keys
        idx = one["index"].to(torch.int64) & 0xFFFFFFFF
        assert bool((idx[1:] > idx[:-1]).all()) and int(idx[-1]) < int(nd.n_distinct)
        # 2. abundance filter and metadata identities (src/main.rs:778: seqlen = last - first + 2; end = last + l)
        assert int((one["abundance"].to(torch.int64) & 0xFFFF).min()) >= a
        assert bool(((one["src_end"] - one["src_start"] - 14 + 2) == (one["seqlen"].to(torch.int64) & 0xFFFFFFFF)).all())
        assert int(one["src_read"].max()) < SHARD_READS
        # 3. canonical keys: key <= reversed key (first differing position decides); every minimizer hash <= hash_bound
        kk = one["keys"].view(n, 35)
        rev = kk.flip(1)
        neq = kk != rev
        first = neq.to(torch.int8).argmax(1)
        rows = torch.arange(n, device="cuda")
        lt = kk[rows, first].to(torch.float64) + (kk[rows, first] < 0) * 2.0**64 < rev[rows, first].to(torch.float64) + (rev[rows, first] < 0) * 2.0**64
        assert bool((lt | ~neq.any(1)).all())
        assert int(kk.min()) >= 0 and int(kk.max()) <= O.hash_bound(0.003)
        # 4. determinism + the ASCII path give the identical table
        m.reset(0)
        m.ingest_device(db, do, SHARD_READS, nb, 0)
        two = _node_tensors(torch, m.finalize_device(), 35)
        for f in one:
            assert torch.equal(one[f], two[f]), f
        del one, two, kk, rev, neq
    # configs[4]: multik sweep on resident sketches (utils/multik:69-78 runs the binary once per k)
    ks = [10, 15, 20, 25, 30, 35, 40]
    with R.Mdbg(ks[0], 12, 0.003, a) as m:
        m.ingest_packed_device(words.data_ptr(), offs.data_ptr(), SHARD_READS, nb, 0)
        st0 = m.stats()
        for k in ks:
            if k != ks[0]:
                m.reset(k)
            nd = m.finalize_device()
            st = m.stats()
            assert st["n_minimizers"] == st0["n_minimizers"] and (st["n_sketch_tile_launches"] >= 1 if k == ks[0] else st["n_sketch_tile_launches"] == 0)      # no re-sketching
            res = _node_tensors(torch, nd, k)
            assert int(nd.n) > 1_000_000 and int(nd.n_distinct) > int(nd.n)
            with R.Mdbg(k, 12, 0.003, a) as f:
                f.ingest_packed_device(words.data_ptr(), offs.data_ptr(), SHARD_READS, nb, 0)
                fresh = _node_tensors(torch, f.finalize_device(), k)
            for name in res:
                assert torch.equal(res[name], fresh[name]), (k, name)
            del res, fresh


def test_config3_whole_genome_streamed_through_one_gpu():
    """BASELINE configs[3] at FULL size on ONE GPU: the eight shards of the synthetic human data set (3 Gb genome, 10.4 M reads, ~156 Gbases)
    generated shard by shard on the device, packed, and pushed through one context as eight batches with global ordinals; then one
    finalize.  The N=1 anchor of the 1 -> 8 curve and the table at ~10^8 distinct keys.  Checked: the invariants of the shard test, that the
    window count is the sum of the shards' (windows never span reads), and that the table grew without losing or duplicating a key (a second
    pass into a table sized up front gives the same node, distinct and window counts and the same key / index / abundance checksums)."""
    import json
    import os
    import time
    import torch
    import rust_mdbg_amd as R
    k, l, d, a, W = 35, 14, 0.003, 2, 8
    free0 = torch.cuda.mem_get_info()[0]
    rec = {"config": {"k": k, "l": l, "density": d, "minabund": a, "genome_len": SHARD_GENOME * W, "reads": SHARD_READS * W, "batches": W}, "batches": []}
    with R.Mdbg(k, l, d, a) as m:
        words = None
        total_bases = 0
        t_ingest = 0.0
        min_free = free0
        caps = []
        for r in range(W):
            db, do, nb = m.synth_reads_device(seed=1, genome_len=SHARD_GENOME * W, n_reads=SHARD_READS, first_read=r * SHARD_READS)
            if words is None:
                words = torch.zeros(int(nb * 1.02) // 32 + 64, dtype=torch.int64, device="cuda")
            assert (nb + 31) // 32 + 2 <= words.numel()
            torch.cuda.synchronize()
            assert m.pack_device(db, nb, words.data_ptr()) == 0
            m.sync()
            t = time.perf_counter()
            m.ingest_packed_device(words.data_ptr(), do, SHARD_READS, nb, r * SHARD_READS)
            m.sync()
            dt = time.perf_counter() - t
            st = m.stats()
            t_ingest += dt
            total_bases += nb
            min_free = min(min_free, torch.cuda.mem_get_info()[0])
            caps.append(st["table_capacity"])
            rec["batches"].append({"bases": nb, "ingest_ms": dt * 1e3, "minimizers_total": st["n_minimizers"], "windows_total": st["n_windows"],
                                   "distinct_total": st["n_distinct"], "table_capacity": st["table_capacity"]})
        t = time.perf_counter()
        nd = m.finalize_device()
        m.sync()
        t_fin = time.perf_counter() - t
        st = m.stats()
        min_free = min(min_free, torch.cuda.mem_get_info()[0])
        n = int(nd.n)
        assert 150e9 < total_bases < 160e9 and st["n_slow_tiles"] == 0 and st["n_reads"] == SHARD_READS * W
        assert n > 8_000_000 and int(nd.n_distinct) > n
        # invariants (as for one shard)
        idx = _dev(torch, nd.index, n, torch.int32).to(torch.int64) & 0xFFFFFFFF
        assert bool((idx[1:] > idx[:-1]).all()) and int(idx[-1]) < int(nd.n_distinct)
        ab = _dev(torch, nd.abundance, n, torch.int16).to(torch.int64) & 0xFFFF
        assert int(ab.min()) >= a
        sl = _dev(torch, nd.seqlen, n, torch.int32).to(torch.int64) & 0xFFFFFFFF
        assert bool(((_dev(torch, nd.src_end, n, torch.int64) - _dev(torch, nd.src_start, n, torch.int64) - l + 2) == sl).all())
        sr = _dev(torch, nd.src_read, n, torch.int64)
        assert int(sr.max()) < SHARD_READS * W and int(sr.min()) >= 0
        seen_shards = torch.unique(sr // SHARD_READS)
        assert seen_shards.numel() == W                                   # every batch is the A-th sighting of some node: ordinals are global
        kk = _dev(torch, nd.keys, n * k, torch.int64).view(n, k)
        assert int(kk.min()) >= 0 and int(kk.max()) <= O.hash_bound(d)
        step = 4_000_000                                                  # canonical orientation, in slices (the flipped copy of all keys would be 4 GB more)
        for s0 in range(0, n, step):
            q = kk[s0:s0 + step]
            rv = q.flip(1)
            neq = q != rv
            first = neq.to(torch.int8).argmax(1)
            rows = torch.arange(q.shape[0], device="cuda")
            assert bool(((q[rows, first] < rv[rows, first]) | ~neq.any(1)).all())      # all values are < 2^63: signed compare is the unsigned one
            del rv, neq, first, rows
        n_windows, n_distinct, n_min = st["n_windows"], st["n_distinct"], st["n_minimizers"]
        growth = sum(1 for i in range(1, len(caps)) if caps[i] != caps[i - 1])
        rec.update(total_bases=total_bases, nodes=n, distinct=n_distinct, windows=n_windows, minimizers=n_min, ingest_ms=t_ingest * 1e3, finalize_ms=t_fin * 1e3,
                   gbases_per_s=total_bases / (t_ingest + t_fin) / 1e9, table_growth_events=growth, table_capacity_final=st["table_capacity"],
                   peak_hbm_gb=(torch.cuda.mem_get_info()[1] - min_free) / 1e9, note="peak_hbm_gb = device memory not free at the lowest point (other users and the library's block cache included); ingest_ms = sketch + windows + table per batch, inputs packed and resident; synth + pack are outside")
        keys_sum = int(kk.sum(dtype=torch.int64))                         # (wraps: a checksum, compared below)
        idx_sum = int(idx.sum()); ab_sum = int(ab.sum())
        del kk, idx, ab, sl, sr
        # second pass: the same eight batches into a table sized UP FRONT (no growth, no rehash) must give the same graph; on the way every
        # shard is also run alone: windows are additive over batches (they never span reads)
        w_sum = 0
        with R.Mdbg(k, l, d, a, table_capacity_hint=n_distinct + 2 * (n_windows // W)) as g:      # the capacity rule counts a batch's windows as possibly new keys
            for r in range(W):
                with R.Mdbg(k, l, d, a) as f:
                    db, do, nb = f.synth_reads_device(seed=1, genome_len=SHARD_GENOME * W, n_reads=SHARD_READS, first_read=r * SHARD_READS)
                    torch.cuda.synchronize()
                    assert f.pack_device(db, nb, words.data_ptr()) == 0
                    f.ingest_packed_device(words.data_ptr(), do, SHARD_READS, nb, r * SHARD_READS)
                    f.sync()
                    stf = f.stats()
                    w_sum += stf["n_windows"]
                    assert stf["n_minimizers"] == rec["batches"][r]["minimizers_total"] - (rec["batches"][r - 1]["minimizers_total"] if r else 0)
                    g.ingest_packed_device(words.data_ptr(), do, SHARD_READS, nb, r * SHARD_READS)
                    g.sync()
                    if r == 0:
                        cap_g = g.stats()["table_capacity"]
            ng = g.finalize_device()
            stg = g.stats()
            assert stg["table_capacity"] == cap_g, "the pre-sized table must not have grown"
            assert int(ng.n) == n and stg["n_distinct"] == n_distinct and stg["n_windows"] == n_windows
            kg = _dev(torch, ng.keys, n * k, torch.int64)
            assert int(kg.sum(dtype=torch.int64)) == keys_sum
            assert int((_dev(torch, ng.index, n, torch.int32).to(torch.int64) & 0xFFFFFFFF).sum()) == idx_sum
            assert int((_dev(torch, ng.abundance, n, torch.int16).to(torch.int64) & 0xFFFF).sum()) == ab_sum
            del kg
    assert w_sum == n_windows
    rec["second_pass_pre_sized_table"] = "equal (nodes, distinct, windows, key / index / abundance checksums)"
    print("FULL_HUMAN " + json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out) and os.access(out, os.W_OK):
        json.dump(rec, open(os.path.join(out, "full_human.json"), "w"), indent=1)
