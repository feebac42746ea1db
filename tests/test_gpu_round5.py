"""Round-5 additions checked on the GPU."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

# the gather of SMALL tiles (gather_multi_kernel: one wave takes several tiles) is chosen whenever a tile expects fewer than 128 records — with the product's 256-lane
# tiles that is any density below ~0.002 —, and until round 5 only the wave-tile experiment exercised it.  Child process: the hooks are read once per process.
CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import rust_mdbg_amd as R
from oracle import oracle as O
rng = np.random.default_rng(%d)
def rnd(n):
    return rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).tobytes()
# several reads per tile (3 - 9 kb), reads that span tiles (70 kb), empty reads, a read of one homopolymer (no minimizer: empty tiles behind it), tiny reads
reads = []
for i in range(260):
    reads.append(rnd(int(rng.integers(3000, 9000))))
    if i %% 37 == 0: reads.append(b"")
    if i %% 53 == 0: reads.append(rnd(70000))
    if i %% 97 == 0: reads.append(b"A" * 90000)
    if i %% 11 == 0: reads.append(rnd(int(rng.integers(1, 30))))
b, o = O.concat_reads(reads)
for (l, d) in ((12, 0.0005), (14, 0.0002), (10, 0.0015)):
    exp = O.sketch(b, o, l, d)
    with R.Mdbg(5, l, d, 1) as m:
        for rep in range(2):                         # the second pass starts with the slab size the first one learnt
            sk = m.sketch(b, o)
            assert np.array_equal(sk["hashes"], exp["hashes"]) and np.array_equal(sk["pos"], exp["pos"]) and np.array_equal(sk["off"], exp["off"]), (l, d, rep)
    assert len(exp["hashes"]) > 300, len(exp["hashes"])
print("GATHER_MULTI_OK", len(b))
"""


def _child(seed, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, seed)], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and "GATHER_MULTI_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_small_tile_gather_at_low_density_equals_oracle():
    _child(5)


@pytest.mark.gpu
def test_small_tile_gather_survives_slab_overflow_and_several_launches():
    # MDBG_SLAB_CAP0=8: every tile with more than 8 minimizers overflows its slab on the first attempt (the gather of that attempt skips it and shifts the tiles behind
    # it: its output is discarded), the batch runs again with slabs sized from the largest count seen; MDBG_SLAB_BUDGET_MB=1: ~8 launches per batch
    _child(6, MDBG_SLAB_CAP0=8, MDBG_SLAB_BUDGET_MB=1)
    _child(7, MDBG_SLAB_BUDGET_MB=1)
    _child(8, MDBG_SLAB_CAP0=8)


@pytest.mark.gpu
def test_default_build_has_one_tile_shape():
    from rust_mdbg_amd import api
    assert api.load_library().mdbg_build_flags() == 0          # no wave tiles (the only build switch left)


@pytest.mark.gpu
def test_finalize_without_the_claim_map_gives_the_same_nodes():
    """finalize starts from the insertion's claim map since round 5 (fin_mark only moves the marks of keys whose first sighting is not their claimer); MDBG_NO_CLAIMS
    keeps the round-4 path (every key's first sighting marked at finalize) alive: the fuzz seeds give the oracle's nodes on it too"""
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import test_gpu_fuzz as F\n"
             "for s in (1, 2, 3, 5, 8, 13, 21): F.test_fuzz_sketch_and_nodes(s)\nprint('NO_CLAIMS_OK')\n" % (ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=dict(os.environ, MDBG_NO_CLAIMS="1"), timeout=900)
    assert r.returncode == 0 and "NO_CLAIMS_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_owner_table_from_a_skewed_first_batch():
    """the measured owner table (window-minimum histogram -> bins dealt out to the ranks) is built from the FIRST round only — a key's owner may not change
    under a live table.  Here the first round is as unrepresentative as it gets: every rank's first batch comes from a 20-kb genome at 2,000x (a few hundred
    distinct window minima: a few hundred of the 65,536 bins hold everything), the second from a 60-Mb genome.  The table must stay EXACT whatever the balance;
    the balance itself is reported (nodes per rank, max / mean).  Measured on MI355X: 1.28 while bins the first round never saw were spread by a hash of their index,
    1.21 with the analytic quantiles of the window-minimum distribution for them, **1.03** since the bins' weights are a blend of the measurement and a model of a
    uniform genome (the enumerated l-mer hashes under the bound and their chance of being a window's minimum), the measurement trusted by the share of the model's mass
    it has seen (dist_api.inc, build_owner_table).  A representative first round: 1.01 - 1.05 as before (tests/test_gpu_dist_scale.py)."""
    import numpy as np
    import rust_mdbg_amd as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_dist_scale as S
    W, k, l, d, A = 4, 21, 12, 0.004, 2
    n0, n1 = 3000, 40000

    def src(rank, rd):
        # -> (seed, genome_len, n_reads, first_read): round 0 = the tiny genome, round 1 = the large one; ordinals keep the ranks' batches apart
        return (11, 20_000, n0, rank * n0) if rd == 0 else (12, 60_000_000, n1, W * n0 + rank * n1)

    def feed(rank, gen, rd):
        seed, glen, n, first = src(rank, rd)
        db, do, nb = gen.synth_reads_device(seed=seed, genome_len=glen, n_reads=n, first_read=first)
        return db, do, n, nb, first
    parts = S._run_ranks(W, k, l, d, A, feed, chunks=2, whole=False, rounds=2)
    with R.Mdbg(k, l, d, A, device=0) as one, R.Mdbg(k, l, d, A, device=0) as gen:
        for rd in range(2):
            for r in range(W):
                seed, glen, n, first = src(r, rd)
                db, do, nb = gen.synth_reads_device(seed=seed, genome_len=glen, n_reads=n, first_read=first)
                one.ingest_device(db, do, n, nb, first)
        ref = one.finalize()
    assert ref["n_nodes"] > 50000
    S._assert_partitions_equal(parts, ref)
    sizes = [p["n"] for p in parts]
    ratio = max(sizes) / (sum(sizes) / W)
    print("nodes per rank %r, max / mean %.3f" % (sizes, ratio))
    assert ratio < 1.10, sizes


@pytest.mark.gpu
@pytest.mark.parametrize("W,chunks,k2", [(2, 1, 27), (4, 2, 15), (3, 3, 31)])
def test_dist_reset_to_another_k_under_segments(W, chunks, k2):
    """mdbg_dist_reset(new k) with the default exchange: the foreign sketches hold only the hashes of the old k's windows, so the library exchanges the rounds again
    for the new k (new owner lists, new segments into the same store regions) without sketching anything again; the partitions put together equal ONE context that
    was reset to the same k — both for a larger and for a smaller k (neither is a subset of the other: the owner is a function of the window's smallest hash)"""
    import ctypes as C
    import threading
    import numpy as np
    import rust_mdbg_amd as R
    from rust_mdbg_amd import api, dist_c
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_dist_scale as S
    from thread_comm import ThreadWorld
    k, l, d, A, n_reads, rounds = 21, 12, 0.004, 2, 6000, 2
    genome = 20_000_000
    L = api.load_library()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(dist_c.Comm), C.POINTER(C.c_int)]
    L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_reset.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    world = ThreadWorld(W)
    parts, traffic, errs = [None] * W, [None] * W, []

    def first_of(rank, rd):
        return (rd * W + rank) * n_reads

    def body(rank):
        try:
            cm, keep = world.comm(rank)
            P = api.Params(k=k, l=l, density=d, min_abundance=A, reads_already_hpc=0, device=0, flags=0, table_capacity_hint=0)
            err = C.c_int()
            h = L.mdbg_dist_create(C.byref(P), C.byref(cm), C.byref(err))
            assert h, err.value
            assert L.mdbg_dist_set_pipeline(h, chunks) == 0
            with R.Mdbg(k, l, d, A, device=0) as gen:
                for rd in range(rounds):
                    idle = rd == rounds - 1 and rank == W - 1          # the last rank sits out the last round
                    db, do, nb = gen.synth_reads_device(seed=5, genome_len=genome, n_reads=n_reads, first_read=first_of(rank, rd))
                    assert L.mdbg_dist_ingest_batch_device(h, db, do, 0 if idle else n_reads, nb, first_of(rank, rd)) == 0
                nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
                assert L.mdbg_dist_finalize(h, C.byref(nd), C.byref(row), C.byref(ng)) == 0
                bi0 = C.c_uint64()
                L.mdbg_dist_traffic(h, C.byref(bi0), None, None)
                world.bar.wait()
                e = L.mdbg_dist_reset(h, k2)
                assert e == 0, e
                assert L.mdbg_dist_finalize(h, C.byref(nd), C.byref(row), C.byref(ng)) == 0
                bi1 = C.c_uint64()
                L.mdbg_dist_traffic(h, C.byref(bi1), None, None)
                n = int(nd.n)
                cp = lambda p, cnt, dt: gen.to_host(C.cast(p, C.c_void_p).value, cnt * np.dtype(dt).itemsize, dt) if cnt else np.empty(0, dt)
                parts[rank] = dict(n=n, ng=int(ng.value), n_distinct=int(nd.n_distinct), n_wrapped=int(nd.n_wrapped), row=cp(row, n, np.uint64),
                                   keys=cp(nd.keys, n * k2, np.uint64).reshape(n, k2), index=cp(nd.index, n, np.uint32), abundance=cp(nd.abundance, n, np.uint16),
                                   seqlen=cp(nd.seqlen, n, np.uint32), shift_full=cp(nd.shift_full, 2 * n, np.uint64).reshape(n, 2), src_read=cp(nd.src_read, n, np.uint64),
                                   src_start=cp(nd.src_start, n, np.uint64), src_end=cp(nd.src_end, n, np.uint64))
                traffic[rank] = (bi0.value, bi1.value - bi0.value)
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
    with R.Mdbg(k, l, d, A, device=0) as one, R.Mdbg(k, l, d, A, device=0) as gen:
        for rd in range(rounds):
            for r in range(W):
                if rd == rounds - 1 and r == W - 1:
                    continue
                db, do, nb = gen.synth_reads_device(seed=5, genome_len=genome, n_reads=n_reads, first_read=first_of(r, rd))
                one.ingest_device(db, do, n_reads, nb, first_of(r, rd))
        one.finalize()
        one.reset(k2)
        ref = one.finalize()
    assert ref["n_nodes"] > 5000
    S._assert_partitions_equal(parts, ref)
    assert all(t[0] > 0 and t[1] > 0 for t in traffic), traffic          # the new k cost an exchange of its own
    print("bytes received per rank, first k / new k:", traffic)


@pytest.mark.gpu
def test_gfa_without_the_sequences_pass_is_the_same_file(tmp_path):
    """run_file(write_sequences=False) copies three columns of the node table to the host instead of all of it (Mdbg.finalize(gfa_only=True)): the .gfa is
    byte for byte the one of the full run, and such a table is refused where the minimizer lists would be needed"""
    import os
    from conftest import GOLDEN
    from rust_mdbg_amd import pipeline
    from rust_mdbg_amd.emit import Emitter
    src = os.path.join(GOLDEN, "reads-0.00.fa.gz")
    full = pipeline.run_file(src, str(tmp_path / "full"), 7, 10, 0.0008, 2, batch_bases=3_000_000)
    light = pipeline.run_file(src, str(tmp_path / "light"), 7, 10, 0.0008, 2, batch_bases=3_000_000, write_sequences=False, threads=4)
    assert open(str(tmp_path / "full.gfa"), "rb").read() == open(str(tmp_path / "light.gfa"), "rb").read()
    assert {f: full[f] for f in full if f != "seconds_until"} == {f: light[f] for f in light if f != "seconds_until"}
    assert not os.path.exists(str(tmp_path / "light.0.sequences"))
    import rust_mdbg_amd as R
    with R.Mdbg(7, 10, 0.0008, 2, device=0) as m:
        reads = [b"ACGTTGCATGCAGTCAGTCGATGCTAGCTAGTCGATCGATGCATGCTAGCATCGATCGATGCATGC" * 40]
        from rust_mdbg_amd.api import concat_reads
        b, o = concat_reads(reads)
        m.ingest(b, o, 0)
        nd = m.finalize(gfa_only=True)
        assert nd["keys"] is None and len(nd["index"]) == nd["n_nodes"] and nd["k"] == 7
        em = Emitter()
        with pytest.raises(ValueError):
            em.edges(nd)
        with pytest.raises(ValueError):
            em.write_sequences(str(tmp_path / "x.sequences"), nd, 10, [(b, o, 0)])


@pytest.mark.gpu
def test_reader_batches_are_page_locked_by_the_first_ingest(tmp_path):
    """mdbg_host_alloc: ordinary memory until an ingest call is given a pointer into it, page-locked from then on (one DMA per batch instead of the staged copy);
    the graph is the one of a run on malloc'd buffers; memory from elsewhere is not touched"""
    import random
    import numpy as np
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    from rust_mdbg_amd.api import load_library
    rnd = random.Random(3)
    genome = bytes(rnd.choice(b"ACGT") for _ in range(300_000))
    with open(str(tmp_path / "r.fa"), "wb") as f:
        for i in range(400):
            a = rnd.randrange(0, len(genome) - 12_000)
            f.write(b">r%d\n" % i + genome[a:a + rnd.randrange(6_000, 12_000)] + b"\n")
    H = load_library()
    out = {}
    for mode in ("malloc", "device-packed", "device-ascii"):
        H.mdbg_release_cached_memory()          # (page-locked allocations are kept for the next reader: start every mode from none)
        with R.Mdbg(5, 8, 0.01, 2, device=0) as m, E.Reader(str(tmp_path / "r.fa"), threads=4, device_buffers=mode != "malloc") as r:
            first, seen = 0, []
            if mode == "device-ascii":
                for b, o in r.batches(600_000, copy=False):
                    assert H.mdbg_host_is_pinned(b.ctypes.data) == (1 if b.ctypes.data in seen else 0)
                    m.ingest(b, o, first)
                    assert H.mdbg_host_is_pinned(b.ctypes.data) == 1
                    seen.append(b.ctypes.data); first += len(o) - 1
            else:
                for pk in r.batches_packed(600_000, copy=False):
                    p = pk["words"].ctypes.data
                    m.ingest_packed(pk, first)
                    assert H.mdbg_host_is_pinned(p) == (0 if mode == "malloc" else 1)
                    seen.append(p); first += len(pk["offsets"]) - 1
            assert len(seen) >= 4 and len(set(seen)) == 2
            nd = m.finalize()
            out[mode] = (nd["n_nodes"], nd["keys"].tobytes(), nd["abundance"].tobytes())
        assert all(H.mdbg_host_is_pinned(p) == 0 for p in seen)          # the reader has given its buffers back
    assert out["malloc"][0] > 100 and out["malloc"] == out["device-packed"] == out["device-ascii"]
    q = H.mdbg_host_alloc(1 << 20)
    assert q and H.mdbg_host_is_pinned(q) == 0
    H.mdbg_host_free(q)
    # an allocation that was page-locked is kept for the next request of about its size, still locked
    with R.Mdbg(5, 8, 0.01, 2, device=0) as m, E.Reader(str(tmp_path / "r.fa"), threads=4, device_buffers=True) as r:
        pk = next(r.batches_packed(600_000, copy=False))
        p1 = pk["words"].ctypes.data
        m.ingest_packed(pk, 0)
        assert H.mdbg_host_is_pinned(p1) == 1
    assert H.mdbg_host_is_pinned(p1) == 0
    with E.Reader(str(tmp_path / "r.fa"), threads=4, device_buffers=True) as r:
        got = {pk["words"].ctypes.data for pk in r.batches_packed(600_000, copy=False)}
        assert p1 in got and H.mdbg_host_is_pinned(p1) == 1          # locked before any ingest call has seen it
    H.mdbg_release_cached_memory()
    with E.Reader(str(tmp_path / "r.fa"), threads=4, device_buffers=True) as r:
        pk = next(r.batches_packed(600_000, copy=False))
        assert H.mdbg_host_is_pinned(pk["words"].ctypes.data) == 0


@pytest.mark.gpu
def test_hip_path_compresses_homopolymers_like_the_references_python_helper():
    """the ACGT cases of tests/golden/reference_py_vectors.json (outputs of the reference's utils/remove_homopoly.py): with density 1 every l-mer of the
    compressed read is selected, so the positions the HIP path reports ARE the compression's run starts — they must spell the helper's output"""
    import json
    import numpy as np
    import rust_mdbg_amd as R
    from conftest import GOLDEN
    v = json.load(open(os.path.join(GOLDEN, "reference_py_vectors.json")))
    cases = [c for c in v["hpc"] if set(c["input"]) <= set("ACGT") and len(c["output"]) >= 2]
    assert len(cases) >= 15
    l = 2
    reads = [c["input"].encode() for c in cases]
    offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), np.uint8)
    with R.Mdbg(3, l, 1.0, 1, device=0) as m:
        sk = m.sketch(bases, offs)
    for i, c in enumerate(cases):
        pos = [int(p) for p in sk["pos"][int(sk["off"][i]):int(sk["off"][i + 1])]]
        out = c["output"]
        assert len(pos) == len(out) - l + 1, c
        assert "".join(c["input"][p] for p in pos) == out[:len(out) - l + 1], c
        assert all(p == 0 or c["input"][p - 1] != c["input"][p] for p in pos)


# the insertion's two stages (csrc/table.hip, upsert_wave): first stage to a fingerprint hit, second stage heads (full comparison) and links (the left neighbour's
# match continued: one value compared).  Child process: the hooks are read once per process.
CHILD_INSERT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import rust_mdbg_amd as R
from oracle import oracle as O
rng = np.random.default_rng(7)
genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=400_000).tobytes()
def reads_of(n, seed):
    r = np.random.default_rng(seed); out = []
    for _ in range(n):
        a = int(r.integers(0, len(genome) - 30_000)); s = bytearray(genome[a:a + int(r.integers(12_000, 30_000))])
        for p in r.integers(0, len(s), size=len(s) // 2500): s[int(p)] = b"ACGT"[int(r.integers(0, 4))]      # a few errors: chains break and start again
        out.append(bytes(s) if r.random() < 0.5 else O.revcomp(bytes(s)))
    return out
k, l, d, A = 9, 8, 0.02, 2
batches = [reads_of(30, 100 + i) for i in range(12)]          # ~1.5x per batch, 18x in all: a later batch's windows repeat keys that ONE earlier read created
exp = O.Graph(k, l, d, A)
first = 0
with R.Mdbg(k, l, d, A, device=0) as m:
    for rs in batches:
        b, o = O.concat_reads(rs)
        exp.ingest(b, o, first); m.ingest(b, o, first); first += len(rs)
    r = exp.finalize(with_edges=False); nd = m.finalize(); st = m.stats()
    assert nd["n_nodes"] == r["n_nodes"] > 2000 and nd["n_nodes_before"] == r["n_nodes_before"]
    for f in ("keys", "index", "abundance", "seqlen", "shift_full", "src_read", "src_start", "src_end", "reversed"):
        assert np.array_equal(np.asarray(nd[f]).reshape(-1), np.asarray(r[f]).reshape(-1)), f
    print("LINKS", st["n_link_matches"], "WINDOWS", st["n_windows"], "DISTINCT", st["n_distinct"])
    # one batch with every copy in it (the copies race for the claim), then the same reads once more (every window a repeat)
    m.reset(0)
    allr = [x for rs in batches for x in rs]
    b, o = O.concat_reads(allr)
    m.ingest(b, o, 0); m.ingest(b, o, len(allr))
    e2 = O.Graph(k, l, d, A); e2.ingest(b, o, 0); e2.ingest(b, o, len(allr)); r2 = e2.finalize(with_edges=False); n2 = m.finalize()
    for f in ("keys", "index", "abundance", "seqlen", "shift_full", "src_read", "src_start", "src_end", "reversed"):
        assert np.array_equal(np.asarray(n2[f]).reshape(-1), np.asarray(r2[f]).reshape(-1)), f
print("INSERT_OK")
"""


def _insert_child(**env):
    e = dict(os.environ, MDBG_COUNT_LINKS="1", **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD_INSERT % ROOT], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and "INSERT_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    f = r.stdout.split("LINKS")[1].split()
    return int(f[0]), int(f[2]), int(f[4])


@pytest.mark.gpu
def test_insertion_confirms_neighbouring_repeats_as_links_and_stays_exact():
    """batches that hold a copy or two of a region: most repeats are confirmed as links of their left neighbour (counted: MDBG_COUNT_LINKS); the node table is the
    oracle's field by field — also without links (MDBG_NO_CHAIN), and with a two-bit fingerprint (MDBG_WEAK_FP), under which most probes meet another key behind
    their fingerprint and heads and links fail into the plain walk"""
    links, windows, distinct = _insert_child()
    assert windows > 20_000 and links > (windows - distinct) // 3, (links, windows, distinct)
    assert _insert_child(MDBG_NO_CHAIN=1)[0] == 0
    weak, _, _ = _insert_child(MDBG_WEAK_FP=1)
    assert weak > 0
    _insert_child(MDBG_WEAK_FP=1, MDBG_NO_CHAIN=1)
    _insert_child(MDBG_WEAK_FP=1, MDBG_POISON=1)
    _insert_child(MDBG_FIN_ONE_PASS=1)          # finalize's marking kernel of the claim-map mode in one pass (the default lists the slots that need work and takes two)
