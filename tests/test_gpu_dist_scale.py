"""The multi-GPU layer behind the C ABI at a size where its machinery is really used: four thread-ranks, 1.5 Gbases of 15-kb reads each at the bench's
parameters (k = 35, l = 12, d = 0.002) — owner lists of millions of windows over thousands of spans, the measured owner table, segments of a few
hashes per window, two pipelined chunks, the position fetch at finalize.  The partitions put together equal the node table of ONE context fed all reads
(examples/mdbg_dist_threads.c makes the same comparison on 300 short reads per rank).  And the rare path through it: k-min-mers whose u16 abundance wrapped,
whose reference sighting has to be recovered from foreign sketches of which only segments are resident."""
import ctypes as C
import hashlib
import random
import threading

import numpy as np
import pytest

from thread_comm import ThreadWorld

pytestmark = pytest.mark.gpu
FIELDS = ("keys", "index", "abundance", "seqlen", "shift_full", "src_read", "src_start", "src_end")


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _run_ranks(W, k, l, d, A, feed, chunks, whole, rounds=1, scheme=0, syncmer_s=0):
    """feed(rank, gen[, round]) -> (d_bases, d_offsets, n_reads, n_bases, first_ordinal) in DEVICE memory (kept alive by the caller; n_reads = 0: the rank
    has nothing for the round); -> list of partitions"""
    from rust_mdbg_amd import api, dist_c
    import rust_mdbg_amd as R
    L = api.load_library()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(dist_c.Comm), C.POINTER(C.c_int)]
    L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_set_exchange.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    world = ThreadWorld(W)
    parts = [None] * W
    errs = []

    def body(rank):
        try:
            cm, keep = world.comm(rank)
            P = api.Params(k=k, l=l, density=d, min_abundance=A, reads_already_hpc=0, device=0, flags=0, table_capacity_hint=0, scheme=scheme, syncmer_s=syncmer_s)
            err = C.c_int()
            h = L.mdbg_dist_create(C.byref(P), C.byref(cm), C.byref(err))
            assert h, err.value
            assert L.mdbg_dist_set_pipeline(h, chunks) == 0 and L.mdbg_dist_set_exchange(h, 1 if whole else 0) == 0
            with R.Mdbg(k, l, d, A, device=0) as gen:
                for rd in range(rounds):
                    db, do, n_reads, nb, first = feed(rank, gen) if rounds == 1 else feed(rank, gen, rd)
                    e = L.mdbg_dist_ingest_batch_device(h, db, do, n_reads, nb, first)
                    assert e == 0, e
                nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
                e = L.mdbg_dist_finalize(h, C.byref(nd), C.byref(row), C.byref(ng))
                assert e == 0, e
                n = int(nd.n)
                cp = lambda p, cnt, dt: gen.to_host(C.cast(p, C.c_void_p).value, cnt * np.dtype(dt).itemsize, dt) if cnt else np.empty(0, dt)
                parts[rank] = dict(n=n, ng=int(ng.value), n_distinct=int(nd.n_distinct), n_wrapped=int(nd.n_wrapped), row=cp(row, n, np.uint64),
                                   keys=cp(nd.keys, n * k, np.uint64).reshape(n, k), index=cp(nd.index, n, np.uint32), abundance=cp(nd.abundance, n, np.uint16),
                                   seqlen=cp(nd.seqlen, n, np.uint32), shift_full=cp(nd.shift_full, 2 * n, np.uint64).reshape(n, 2), src_read=cp(nd.src_read, n, np.uint64),
                                   src_start=cp(nd.src_start, n, np.uint64), src_end=cp(nd.src_end, n, np.uint64))
            world.bar.wait()
            L.mdbg_dist_destroy(h)
        except BaseException as ex:          # noqa: BLE001 (a failing rank must not leave the others at a barrier)
            errs.append(ex)
            world.bar.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return parts


def _assert_partitions_equal(parts, ref):
    n_ref = ref["n_nodes"]
    assert sum(p["n"] for p in parts) == n_ref and all(p["ng"] == n_ref and p["n_distinct"] == ref["n_nodes_before"] for p in parts)
    rows = np.concatenate([p["row"] for p in parts])
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(rows[order], np.arange(n_ref, dtype=np.uint64))          # every row of the global table exactly once
    for f in FIELDS:
        got = np.concatenate([p[f] for p in parts])[order]
        assert _digest(got) == _digest(ref[f]), f


@pytest.mark.parametrize("whole", [False, True])
def test_four_ranks_at_bench_parameters_equal_one_context(whole):
    import rust_mdbg_amd as R
    W, k, l, d, n_reads = 4, 35, 12, 0.002, 100000
    genome = 30_000_000 * W

    def feed(rank, gen):
        db, do, nb = gen.synth_reads_device(seed=3, genome_len=genome, n_reads=n_reads, first_read=rank * n_reads)
        return db, do, n_reads, nb, rank * n_reads
    parts = _run_ranks(W, k, l, d, 2, feed, chunks=2, whole=whole)
    with R.Mdbg(k, l, d, 2, device=0) as one, R.Mdbg(k, l, d, 2, device=0) as gen:          # ONE context, the same reads under the same ordinals
        for r in range(W):
            db, do, nb = gen.synth_reads_device(seed=3, genome_len=genome, n_reads=n_reads, first_read=r * n_reads)
            one.ingest_device(db, do, n_reads, nb, r * n_reads)
        ref = one.finalize()
    assert ref["n_nodes"] > 100000
    _assert_partitions_equal(parts, ref)
    sizes = [p["n"] for p in parts]          # the measured owner table keeps the partitions within a few per cent
    assert max(sizes) < 1.05 * (sum(sizes) / W), sizes


def test_eight_ranks_thin_lists_equal_one_context():
    """eight ranks at the human parameters (k = 35 l = 14 d = 0.003): a rank's share of a peer's sketch is ~130 windows per span of 2,048 — the per-entry insertion
    (insert_listed_entries_kernel, several batches per launch), segment passes over many 4,096-entry blocks with bucket boundaries inside them, and the claim map gathered
    across eight batches' boundaries (claims_to_bits_kernel) — against ONE context that sees the same reads under the same ordinals"""
    import rust_mdbg_amd as R
    W, k, l, d, n_reads = 8, 35, 14, 0.003, 30000
    genome = 12_000_000 * W

    def feed(rank, gen):
        db, do, nb = gen.synth_reads_device(seed=11, genome_len=genome, n_reads=n_reads, first_read=rank * n_reads)
        return db, do, n_reads, nb, rank * n_reads
    parts = _run_ranks(W, k, l, d, 2, feed, chunks=2, whole=False)
    with R.Mdbg(k, l, d, 2, device=0) as one, R.Mdbg(k, l, d, 2, device=0) as gen:
        for r in range(W):
            db, do, nb = gen.synth_reads_device(seed=11, genome_len=genome, n_reads=n_reads, first_read=r * n_reads)
            one.ingest_device(db, do, n_reads, nb, r * n_reads)
        ref = one.finalize()
    assert ref["n_nodes"] > 50000
    _assert_partitions_equal(parts, ref)


@pytest.mark.parametrize("whole,chunks", [(False, 1), (False, 3), (True, 1)])
def test_wrapped_abundances_across_ranks(whole, chunks):
    """every k-min-mer of a repeated unit occurs ~84,000 times, spread over the reads of both ranks: the reference's u16 abundance wraps and its entry describes
    sighting A + 65536 * floor((count - A) / 65536) (src/main.rs:676-684), which the owner recovers by re-scanning the windows — its own sketch whole, the
    peer's through the window list (only segments of it are resident)"""
    import torch
    import rust_mdbg_amd as R
    rnd = random.Random(9)
    unit = bytes(rnd.choice(b"ACGT") for _ in range(190))
    reads = []
    for _ in range(420):
        a = rnd.randrange(190)
        reads.append((unit * 202)[a:a + 190 * 200])
    k, l, d, A, W = 3, 8, 0.05, 2, 2
    split = [0, 200, 420]
    keep = {}

    def feed(rank, gen):
        rs = reads[split[rank]:split[rank + 1]]
        offs = np.zeros(len(rs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in rs])
        tb = torch.from_numpy(np.frombuffer(b"".join(rs), dtype=np.uint8).copy()).cuda()
        to = torch.from_numpy(offs.view(np.int64)).cuda()
        torch.cuda.synchronize()
        keep[rank] = (tb, to)
        return tb.data_ptr(), to.data_ptr(), len(rs), int(offs[-1]), split[rank]
    parts = _run_ranks(W, k, l, d, A, feed, chunks=chunks, whole=whole)
    with R.Mdbg(k, l, d, A, device=0) as one:
        one.ingest_reads(reads[:200], 0)
        one.ingest_reads(reads[200:], 200)
        ref = one.finalize()
    assert ref["n_nodes"] >= 3 and ref["n_wrapped"] >= ref["n_nodes"] - 2 and int(np.max(ref["src_read"])) > 200
    _assert_partitions_equal(parts, ref)
    assert sum(p["n_wrapped"] for p in parts) == ref["n_wrapped"]


@pytest.mark.parametrize("chunks", [1, 2])
def test_syncmer_scheme_and_uneven_rounds_across_three_ranks(chunks):
    """the syncmer scheme through the multi-GPU layer (its hash bound is density * 4^l: the owner thresholds and the measured table are built from that), three
    rounds per rank of which one rank sits out the second (n_reads = 0) and another the third"""
    import rust_mdbg_amd as R
    W, k, l, s_, d, per = 3, 10, 12, 4, 0.05, 1500
    genome = 6_000_000

    def first_of(rank, rd):
        return (rd * W + rank) * per

    def feed(rank, gen, rd):
        if (rank, rd) in ((1, 1), (2, 2)):
            return None, None, 0, 0, first_of(rank, rd)
        db, do, nb = gen.synth_reads_device(seed=5, genome_len=genome, n_reads=per, mean_len=9000, sd_len=1500, min_len=2000, max_len=16000, err_ppm=2000, first_read=first_of(rank, rd))
        return db, do, per, nb, first_of(rank, rd)
    parts = _run_ranks(W, k, l, d, 2, feed, chunks=chunks, whole=False, rounds=3, scheme=1, syncmer_s=s_)
    with R.Mdbg(k, l, d, 2, device=0, syncmer_s=s_) as one, R.Mdbg(k, l, d, 2, device=0) as gen:
        for rd in range(3):
            for r in range(W):
                if (r, rd) in ((1, 1), (2, 2)):
                    continue
                db, do, nb = gen.synth_reads_device(seed=5, genome_len=genome, n_reads=per, mean_len=9000, sd_len=1500, min_len=2000, max_len=16000, err_ppm=2000, first_read=first_of(r, rd))
                one.ingest_device(db, do, per, nb, first_of(r, rd))
        ref = one.finalize()
    assert ref["n_nodes"] > 1000
    _assert_partitions_equal(parts, ref)


def test_ranks_that_disagree_about_the_exchange_are_told_so():
    """every rank must run the same library version, exchange mode, chunk count and sketch parameters; the first round compares them and fails on every rank
    (MDBG_E_PARAM) instead of hanging in a mismatched exchange"""
    from rust_mdbg_amd import api, dist_c
    import rust_mdbg_amd as R
    L = api.load_library()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(dist_c.Comm), C.POINTER(C.c_int)]
    L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mdbg_dist_set_exchange.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    W = 2
    world = ThreadWorld(W)
    rc = [None] * W
    errs = []

    def body(rank):
        try:
            cm, keep = world.comm(rank)
            P = api.Params(k=9, l=12, density=0.004, min_abundance=2, reads_already_hpc=0, device=0, flags=0, table_capacity_hint=0)
            err = C.c_int()
            h = L.mdbg_dist_create(C.byref(P), C.byref(cm), C.byref(err))
            assert h, err.value
            assert L.mdbg_dist_set_exchange(h, rank) == 0            # rank 0: segments, rank 1: whole sketches
            with R.Mdbg(9, 12, 0.004, 2, device=0) as gen:
                db, do, nb = gen.synth_reads_device(seed=2, genome_len=150000, n_reads=100, mean_len=9000, sd_len=1500, min_len=2000, max_len=16000, first_read=rank * 100)
                rc[rank] = L.mdbg_dist_ingest_batch_device(h, db, do, 100, nb, rank * 100)
            world.bar.wait()
            L.mdbg_dist_destroy(h)
        except BaseException as ex:          # noqa: BLE001
            errs.append(ex)
            world.bar.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    assert rc == [-1, -1], rc          # MDBG_E_PARAM on both
