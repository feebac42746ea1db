"""The multi-GPU layer behind the C ABI at a size where its machinery is really used: four thread-ranks, 1.5 Gbases of 15-kb reads each at the bench's
parameters (k = 35, l = 12, d = 0.002) — owner lists of millions of windows over thousands of spans, the measured owner table, segments of a few
hashes per window, two pipelined chunks, the position fetch at finalize.  The partitions put together equal the node table of ONE context fed all reads
(examples/mdbg_dist_threads.c makes the same comparison on 300 short reads per rank)."""
import ctypes as C
import hashlib
import threading

import numpy as np
import pytest

from thread_comm import ThreadWorld

pytestmark = pytest.mark.gpu


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("whole", [False, True])
def test_four_ranks_at_bench_parameters_equal_one_context(whole):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import api, dist_c
    W, k, l, d, n_reads = 4, 35, 12, 0.002, 100000
    genome = 30_000_000 * W
    L = api.load_library()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(dist_c.Comm), C.POINTER(C.c_int)]
    L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_set_exchange.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    L.mdbg_dist_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    world = ThreadWorld(W)
    parts = [None] * W
    errs = []

    def body(rank):
        try:
            cm, keep = world.comm(rank)
            P = api.Params(k=k, l=l, density=d, min_abundance=2, reads_already_hpc=0, device=0, flags=0, table_capacity_hint=0)
            err = C.c_int()
            h = L.mdbg_dist_create(C.byref(P), C.byref(cm), C.byref(err))
            assert h, err.value
            assert L.mdbg_dist_set_pipeline(h, 2) == 0 and L.mdbg_dist_set_exchange(h, 1 if whole else 0) == 0
            with R.Mdbg(k, l, d, 2, device=0) as gen:
                db, do, nb = gen.synth_reads_device(seed=3, genome_len=genome, n_reads=n_reads, first_read=rank * n_reads)
                e = L.mdbg_dist_ingest_batch_device(h, db, do, n_reads, nb, rank * n_reads)
                assert e == 0, e
                nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
                e = L.mdbg_dist_finalize(h, C.byref(nd), C.byref(row), C.byref(ng))
                assert e == 0, e
                n = int(nd.n)
                cp = lambda p, cnt, dt: gen.to_host(C.cast(p, C.c_void_p).value, cnt * np.dtype(dt).itemsize, dt) if cnt else np.empty(0, dt)
                a, b, q = C.c_uint64(), C.c_uint64(), C.c_uint64()
                L.mdbg_dist_traffic(h, C.byref(a), C.byref(b), C.byref(q))
                parts[rank] = dict(n=n, ng=int(ng.value), n_distinct=int(nd.n_distinct), row=cp(row, n, np.uint64), keys=cp(nd.keys, n * k, np.uint64).reshape(n, k),
                                   index=cp(nd.index, n, np.uint32), abundance=cp(nd.abundance, n, np.uint16), seqlen=cp(nd.seqlen, n, np.uint32),
                                   shift_full=cp(nd.shift_full, 2 * n, np.uint64).reshape(n, 2), src_read=cp(nd.src_read, n, np.uint64),
                                   src_start=cp(nd.src_start, n, np.uint64), src_end=cp(nd.src_end, n, np.uint64), bytes_in=int(a.value))
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
    # ONE context, the same reads under the same ordinals
    with R.Mdbg(k, l, d, 2, device=0) as one, R.Mdbg(k, l, d, 2, device=0) as gen:
        for r in range(W):
            db, do, nb = gen.synth_reads_device(seed=3, genome_len=genome, n_reads=n_reads, first_read=r * n_reads)
            one.ingest_device(db, do, n_reads, nb, r * n_reads)
        ref = one.finalize()
    n_ref = ref["n_nodes"]
    assert n_ref > 100000 and sum(p["n"] for p in parts) == n_ref and all(p["ng"] == n_ref and p["n_distinct"] == ref["n_nodes_before"] for p in parts)
    rows = np.concatenate([p["row"] for p in parts])
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(rows[order], np.arange(n_ref, dtype=np.uint64))          # every row of the global table exactly once
    for f in ("keys", "index", "abundance", "seqlen", "shift_full", "src_read", "src_start", "src_end"):
        got = np.concatenate([p[f] for p in parts])[order]
        assert _digest(got) == _digest(ref[f]), f
    # balance and volume: the measured owner table keeps the partitions within a few per cent; segments move well under what whole sketches would
    sizes = [p["n"] for p in parts]
    assert max(sizes) < 1.05 * (sum(sizes) / W), sizes
