"""GPU parity: libmdbg_hip.so (through its C ABI) against the CPU oracle, bit-exact.

Every test here needs a real MI355X (`-m gpu`).  Inputs are seeded; sizes are chosen so the oracle finishes in
seconds.  Edge cases follow SURVEY.md §8a-Q: empty / short / ragged reads, N, bad bytes, long homopolymers,
reads straddling tile boundaries, strict `> k`, palindromes, batch splitting and ordering, multi-k.
"""
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _mdbg():
    import rust_mdbg_amd as R
    return R


def rand_reads(seed, n, lo, hi, alphabet=b"ACGT", hp=0.0):
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        ln = rnd.randint(lo, hi)
        s = bytearray()
        while len(s) < ln:
            c = rnd.choice(alphabet)
            run = 1
            if hp and rnd.random() < hp:
                run = rnd.randint(2, 40)
            s.extend([c] * run)
        out.append(bytes(s[:ln]))
    return out


def oracle_graph(reads, k, l, d, A, hpc=False, splits=None):
    g = O.Graph(k, l, d, A, already_hpc=hpc)
    b, o = O.concat_reads(reads)
    assert g.ingest(b, o) == 0
    return g.finalize(with_edges=False)


def assert_sketch_equal(got, exp):
    assert np.array_equal(got["off"], exp["off"])
    assert np.array_equal(got["hashes"], exp["hashes"])
    assert np.array_equal(got["pos"], exp["pos"])


NODE_FIELDS = ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed")


def assert_nodes_equal(got, exp):
    assert got["n_nodes_before"] == exp["n_nodes_before"]
    assert got["n_nodes"] == exp["n_nodes"]
    for f in NODE_FIELDS:
        assert np.array_equal(got[f], exp[f]), f


def run_gpu(reads, k, l, d, A, hpc=False, flags=0, batches=None, hint=0):
    R = _mdbg()
    with R.Mdbg(k, l, d, A, reads_already_hpc=hpc, flags=flags, table_capacity_hint=hint) as m:
        if batches is None:
            m.ingest_reads(reads, 0)
        else:
            for (lo, hi) in batches:
                m.ingest_reads(reads[lo:hi], lo)
        return m.finalize(), m.stats()


# ------------------------------------------------------------------------------------------------------
def test_example_cfg1(example_reads):
    """BASELINE.json configs[0]: example/reads-0.00.fa.gz, k=7 l=10 d=0.0008 minabund=2"""
    gold = json.load(open(os.path.join(GOLDEN, "example_cfg1.json")))
    z = np.load(os.path.join(GOLDEN, "example_cfg1_nodes.npz"))
    c = gold["config"]
    R = _mdbg()
    b, o = O.concat_reads(example_reads)
    with R.Mdbg(c["k"], c["l"], c["density"], c["minabund"]) as m:
        sk = m.sketch(b, o)
        assert len(sk["hashes"]) == gold["n_minimizers"]
        assert_sketch_equal(sk, dict(hashes=z["hashes"], pos=z["pos"], off=z["off"]))
        m.ingest(b, o, 0)
        r = m.finalize()
        st = m.stats()
    assert st["n_windows"] == gold["n_windows"] and st["n_minimizers"] == gold["n_minimizers"]
    assert r["n_nodes"] == gold["n_nodes"] and r["n_nodes_before"] == gold["n_nodes_before"]
    for f in ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed"):
        assert np.array_equal(r[f], z[f]), f
    assert st["n_slow_tiles"] == 0


@pytest.mark.parametrize("l,d", [(10, 0.0008), (12, 0.003), (14, 0.003), (12, 0.002), (5, 0.01), (2, 0.02), (13, 0.02)])
@pytest.mark.parametrize("hpc", [False, True])
def test_sketch_random_reads(l, d, hpc):
    reads = rand_reads(100 + l, 40, 0, 60000, hp=0.05)
    reads += [b"", b"A", b"ACGT" * 2, b"", b"C" * 500, rand_reads(5, 1, 200000, 200000)[0], b""]
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, d, hpc)
    assert exp["err"] == 0
    R = _mdbg()
    with R.Mdbg(5, l, d, 2, reads_already_hpc=hpc) as m:
        assert_sketch_equal(m.sketch(b, o), exp)
    with R.Mdbg(5, l, d, 2, reads_already_hpc=hpc, flags=1) as m:      # generic exact kernel on every tile
        assert_sketch_equal(m.sketch(b, o), exp)


@pytest.mark.parametrize("l", [15, 20, 31, 32])
def test_sketch_long_l_takes_generic_path(l):
    reads = rand_reads(l, 10, 100, 30000)
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, 0.01)
    R = _mdbg()
    with R.Mdbg(5, l, 0.01, 2) as m:
        assert_sketch_equal(m.sketch(b, o), exp)


def test_sketch_dense_density_overflows_to_generic():
    """default density of the reference CLI is 0.10: far more candidates than a slab holds"""
    reads = rand_reads(9, 6, 50000, 90000)
    b, o = O.concat_reads(reads)
    for d in (0.1, 0.5, 1.0):
        exp = O.sketch(b, o, 12, d)
        R = _mdbg()
        with R.Mdbg(10, 12, d, 2) as m:
            assert_sketch_equal(m.sketch(b, o), exp)


def test_sketch_low_complexity_and_long_homopolymers():
    rnd = random.Random(4)
    core = rand_reads(1, 1, 3000, 3000)[0]
    reads = [
        b"AC" * 40000,                                                   # every l-mer has one of two hashes
        core + b"A" * 100 + core,                                        # homopolymer shorter than the halo
        core + b"T" * 5000 + core[:700] + b"G" * 70000 + core,           # homopolymers longer than a tile
        b"G" * 300000 + core,                                            # read starts with a 300 kb run
        (b"ACGTTGCA" * 3 + b"C" * 200) * 300,
        bytes(rnd.choice(b"AC") for _ in range(100000)),
    ]
    b, o = O.concat_reads(reads)
    for l, d in ((12, 0.003), (10, 0.05), (14, 0.3)):
        exp = O.sketch(b, o, l, d)
        R = _mdbg()
        with R.Mdbg(5, l, d, 2) as m:
            assert_sketch_equal(m.sketch(b, o), exp)


def test_sketch_with_N():
    rnd = random.Random(8)
    base = rand_reads(2, 8, 20000, 90000)
    reads = []
    for i, r in enumerate(base):
        r = bytearray(r)
        for _ in range(i):                                               # read 0 stays clean
            p = rnd.randrange(len(r))
            r[p:p + rnd.choice([1, 1, 2, 30])] = b"N" * rnd.choice([1, 1, 2, 30])
        reads.append(bytes(r))
    reads += [b"N" * 1000, b"ACGTN" * 5000, b"N" + base[0][:5000] + b"N",
              base[1][:3000] + b"N" + b"A" * 400 + base[2][:3000],      # N then a homopolymer longer than the halo
              base[1][:70000] + b"N" * 200000 + base[2][:70000]]        # N gap spanning whole tiles
    b, o = O.concat_reads(reads)
    for l, d in ((12, 0.003), (10, 0.02)):
        exp = O.sketch(b, o, l, d)
        assert exp["err"] == 0
        R = _mdbg()
        with R.Mdbg(5, l, d, 2) as m:
            assert_sketch_equal(m.sketch(b, o), exp)


def test_alphabet_error_rule():
    """reference: nthash panics iff a read with HPC length >= l holds a byte outside ACGTN"""
    R = _mdbg()
    good = rand_reads(3, 3, 5000, 6000)
    for bad, is_err in ((b"ACGTacgtACGTACGTAC", True), (b"AAAAAAAAAAAAAAAAAAAAx", False), (b"xxxxxxxxxxxxxxxx", True),
                        (good[0][:2000] + b"R" + good[0][2000:], True), (b"ACGTx", False)):
        reads = [good[1], bad, good[2]]
        b, o = O.concat_reads(reads)
        exp = O.sketch(b, o, 12, 0.01)
        assert (exp["err"] != 0) == is_err
        with R.Mdbg(5, 12, 0.01, 2) as m:
            if is_err:
                with pytest.raises(R.MdbgError) as ei:
                    m.sketch(b, o)
                assert ei.value.code == -2
                # sketch_only does not poison the context; ingest does
                with pytest.raises(R.MdbgError):
                    m.ingest(b, o, 0)
                with pytest.raises(R.MdbgError) as e2:
                    m.ingest_reads(good, 10)
                assert e2.value.code == -6
            else:
                assert_sketch_equal(m.sketch(b, o), exp)


def test_many_tiny_reads_and_tile_straddling():
    rnd = random.Random(11)
    reads = []
    for _ in range(30000):
        reads.append(bytes(rnd.choice(b"ACGT") for _ in range(rnd.choice([0, 1, 2, 11, 12, 13, 30, 31, 80]))))
    reads += rand_reads(12, 5, 65530, 65545)                             # ends right at tile boundaries
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, 12, 0.02)
    R = _mdbg()
    with R.Mdbg(3, 12, 0.02, 1) as m:
        assert_sketch_equal(m.sketch(b, o), exp)
        m.ingest(b, o)
        assert_nodes_equal(m.finalize(), oracle_graph(reads, 3, 12, 0.02, 1))


@pytest.mark.parametrize("k,l,d,A", [(7, 10, 0.0008, 2), (21, 12, 0.003, 2), (35, 12, 0.002, 2), (5, 12, 0.01, 1), (3, 8, 0.02, 3),
                                     (4, 12, 0.01, 4), (10, 12, 0.003, 8), (2, 12, 0.003, 2)])
def test_graph_synthetic_reads(k, l, d, A):
    """HiFi-shaped reads (0.1 % errors, both strands) over a small genome: 30-50x coverage"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(k * 100 + l, 400000, 900, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000)
    exp = oracle_graph(reads, k, l, d, A)
    got, st = run_gpu(reads, k, l, d, A)
    assert st["n_windows"] == exp["n_windows"] and st["n_minimizers"] == exp["n_minimizers"]
    assert_nodes_equal(got, exp)
    assert exp["n_nodes"] > 100


def test_graph_skiphpc_and_generic_path_agree():
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(77, 200000, 300, mean_len=12000, sd_len=1500, min_len=3000, max_len=20000, err_ppm=2000)
    exp = oracle_graph(reads, 9, 12, 0.004, 2, hpc=True)
    got, _ = run_gpu(reads, 9, 12, 0.004, 2, hpc=True)
    assert_nodes_equal(got, exp)
    got2, st2 = run_gpu(reads, 9, 12, 0.004, 2, hpc=True, flags=1)
    assert_nodes_equal(got2, exp)
    assert st2["n_slow_tiles"] == st2["n_tiles"] > 0


def test_batch_split_and_order_invariance():
    """results depend on first_read_ordinal only, not on batch boundaries or call order"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(5, 150000, 400, mean_len=9000, sd_len=2000, min_len=1000, max_len=20000, err_ppm=1500)
    k, l, d, A = 8, 12, 0.004, 2
    exp = oracle_graph(reads, k, l, d, A)
    one, _ = run_gpu(reads, k, l, d, A)
    assert_nodes_equal(one, exp)
    split, _ = run_gpu(reads, k, l, d, A, batches=[(0, 1), (1, 130), (130, 131), (131, 400)])
    assert_nodes_equal(split, exp)
    shuffled, _ = run_gpu(reads, k, l, d, A, batches=[(300, 400), (0, 50), (120, 300), (50, 120)])
    assert_nodes_equal(shuffled, exp)
    # tiny initial table: forces several grow-and-rehash rounds
    grown, st = run_gpu(reads, k, l, d, A, batches=[(i, i + 40) for i in range(0, 400, 40)])
    assert_nodes_equal(grown, exp)


def test_dense_rounds_are_inserted_in_slices(monkeypatch):
    """rounds with very many window starts (dense settings) go into the table slice by slice, each checked on the device against the keys the table
    holds by then; the table grows when a slice does not fit and the round resumes there (csrc/api.inc, insert_resident_impl).  MDBG_INSERT_SLICE
    shrinks the slice so that a small input takes that path: several slices, several growths, one and several batches — same table as the oracle"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(11, 120000, 300, mean_len=9000, sd_len=2000, min_len=1000, max_len=20000, err_ppm=1500)
    k, l, d, A = 10, 12, 0.05, 2
    exp = oracle_graph(reads, k, l, d, A)
    plain, st0 = run_gpu(reads, k, l, d, A)
    assert_nodes_equal(plain, exp)
    monkeypatch.setenv("MDBG_INSERT_SLICE", "4096")
    sliced, st1 = run_gpu(reads, k, l, d, A)
    assert_nodes_equal(sliced, exp)
    assert st1["n_windows"] == st0["n_windows"] and st1["n_distinct"] == st0["n_distinct"]
    assert st1["table_capacity"] < st0["table_capacity"]                 # sized for the keys met so far, not for every window of the round
    batched, _ = run_gpu(reads, k, l, d, A, batches=[(0, 100), (100, 101), (101, 300)])
    assert_nodes_equal(batched, exp)


def test_second_batch_overflows_its_slabs_behind_a_speculative_insertion():
    """from the second batch on the window count, capacity check and insertion are launched BEHIND the batch's sketch, before the host has looked at
    it (csrc/api.inc, sketch_device_impl `fused`).  A batch whose tiles hold far more minimizers than the density predicts makes the sketch run again
    with larger slabs: the speculative insertion must have done nothing, the repeated round must insert exactly once.  Graph == oracle."""
    from rust_mdbg_amd import synth
    plain = synth.synth_reads(21, 100000, 120, mean_len=9000, sd_len=2000, min_len=1000, max_len=20000, err_ppm=1000)
    k, l, d, A = 6, 12, 0.004, 2
    R = _mdbg()
    b0, o0 = O.concat_reads(plain)
    sk = O.sketch(b0, o0, l, d)
    # an l-mer of the data whose hash is below the threshold, repeated: every 12th position of such a read is a minimizer (25 x the density)
    r0 = next(r for r in range(len(plain)) if sk["off"][r + 1] > sk["off"][r])
    p = int(sk["pos"][sk["off"][r0]])
    unit = bytes(plain[r0][p:p + 400])
    hp = bytearray()
    for c in unit:                       # the l-mer as the sketch saw it: homopolymer-compressed
        if not hp or hp[-1] != c:
            hp.append(c)
    unit = bytes(hp[:l])
    assert len(unit) == l and unit[0] != unit[-1]
    dense = [unit * 6000, unit * 9000, plain[3] + unit * 5000 + plain[4]]
    reads = plain + dense
    exp = oracle_graph(reads, k, l, d, A)
    got, st = run_gpu(reads, k, l, d, A, batches=[(0, len(plain)), (len(plain), len(reads))])
    assert_nodes_equal(got, exp)
    three, _ = run_gpu(reads, k, l, d, A, batches=[(0, 60), (60, len(plain) + 1), (len(plain) + 1, len(reads))])
    assert_nodes_equal(three, exp)


def test_strict_k_and_palindromes():
    rnd = random.Random(21)
    s = bytes(rnd.choice(b"ACGT") for _ in range(6000))
    l, d = 8, 0.02
    b, o = O.concat_reads([s])
    m_exp = len(O.sketch(b, o, l, d)["hashes"])
    R = _mdbg()
    for k in (m_exp - 1, m_exp, m_exp + 1):
        got, st = run_gpu([s], k, l, d, 1)
        exp = oracle_graph([s], k, l, d, 1)
        assert st["n_windows"] == exp["n_windows"] == (2 if k == m_exp - 1 else 0)
        assert_nodes_equal(got, exp)
    # a read followed by its reverse complement sketches to the mirrored minimizer list: with k=2..3 windows of the
    # junction are palindromic k-min-mers (window == reversed window): normalize() must report reversed=true
    rc = O.revcomp(s)
    n_pal = 0
    for k in (2, 3, 4):
        reads = [s + rc, rc + s]
        exp = oracle_graph(reads, k, l, d, 1)
        got, _ = run_gpu(reads, k, l, d, 1)
        assert_nodes_equal(got, exp)
        pal = [i for i in range(exp["n_nodes"]) if list(exp["keys"][i]) == list(exp["keys"][i][::-1])]
        assert all(exp["reversed"][i] == 1 for i in pal)
        n_pal += len(pal)
    assert n_pal > 0


def test_multik_reset_reuses_resident_sketches():
    """utils/multik: same reads, k = 10, 15, 20 ... — sketches stay resident, the table is cleared and refilled"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(31, 200000, 300, mean_len=14000, sd_len=1500, min_len=5000, max_len=20000, err_ppm=1000)
    l, d, A = 12, 0.003, 2
    R = _mdbg()
    with R.Mdbg(10, l, d, A) as m:
        m.ingest_reads(reads, 0)
        for k in (10, 15, 20, 25, 10):
            if k != m.k or k == 10:
                m.reset(k)
            got = m.finalize()
            assert_nodes_equal(got, oracle_graph(reads, k, l, d, A))
        m.reset(0)                                                        # drop everything
        assert m.stats()["n_minimizers"] == 0
        m.ingest_reads(reads[:50], 0)
        assert_nodes_equal(m.finalize(), oracle_graph(reads[:50], 10, l, d, A))


def test_concurrent_ingest_from_host_threads():
    """mdbg_ingest_batch from several host threads at once (the reference calls process_read_aux from --threads workers,
    src/main.rs:834-913): the node table must not depend on the interleaving, only on the read ordinals."""
    import threading
    reads = rand_reads(41, 300, 2000, 12000)
    reads += [r[50:] for r in reads[:200]]
    k, l, d, A = 6, 12, 0.005, 2
    exp = oracle_graph(reads, k, l, d, A)
    R = _mdbg()
    step = 25
    jobs = [(lo, min(len(reads), lo + step)) for lo in range(0, len(reads), step)]
    for trial in range(3):
        random.Random(trial).shuffle(jobs)
        errs = []
        with R.Mdbg(k, l, d, A) as m:
            def work(js):
                try:
                    for lo, hi in js:
                        m.ingest_reads(reads[lo:hi], lo)
                except BaseException as e:       # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=work, args=(jobs[i::4],)) for i in range(4)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not errs, errs
            got = m.finalize()
            assert m.stats()["n_reads"] == len(reads)
        assert_nodes_equal(got, exp)


def _repeat_reads(seed, n_reads, unit_len, n_units):
    rnd = random.Random(seed)
    unit = bytes(rnd.choice(b"ACGT") for _ in range(unit_len))
    reads = []
    for r in range(n_reads):
        a = rnd.randrange(unit_len)
        reads.append((unit * (n_units + 2))[a:a + unit_len * n_units])
    return reads


@pytest.mark.parametrize("A", [1, 2, 3])
def test_abundance_wrap_keeps_the_reference_sighting(A):
    """k-min-mers seen more than 65536 + A times: the reference's u16 abundance wraps (release build) and the entry is
    refreshed at every sighting whose previous abundance equals A - 1 (src/main.rs:676-684), so seqlen / shift / the sequence
    origin are those of sighting A + 65536 * floor((count - A) / 65536), and the filter sees the wrapped abundance"""
    reads = _repeat_reads(7 + A, 420, 190, 200)          # every k-min-mer of the unit occurs ~84,000 times
    k, l, d = 3, 8, 0.05
    exp = oracle_graph(reads, k, l, d, A)
    got, st = run_gpu(reads, k, l, d, A, batches=[(0, 100), (100, 420)])
    assert exp["n_nodes"] >= 3 and int(np.max(exp["abundance"])) < 65536
    assert_nodes_equal(got, exp)
    assert got["n_wrapped"] >= exp["n_nodes"] - 2          # (almost) every node of this input wrapped
    # the wrapped sighting is NOT the A-th one: the origin lies far into the input
    assert int(np.max(exp["src_read"])) > 200


def test_sparse_density_shift_truncates_to_u16():
    """minimizers more than 65535 bases apart: DbgEntry.shift is stored as u16 (src/main.rs:675) while the .sequences line
    prints the un-truncated value (main.rs:702)"""
    reads = rand_reads(77, 6, 900000, 1200000)
    reads += [reads[0][1000:], reads[1][:-500]]
    k, l, d, A = 2, 12, 0.00002, 1
    exp = oracle_graph(reads, k, l, d, A)
    got, st = run_gpu(reads, k, l, d, A)
    assert_nodes_equal(got, exp)
    sf = np.asarray(exp["shift_full"]).reshape(-1, 2)
    assert exp["n_nodes"] > 20 and int(sf.max()) > 65535
    assert np.array_equal(np.asarray(got["shift"]).reshape(-1, 2), (sf & 0xFFFF).astype(np.uint16))


def test_finalize_is_repeatable_and_incremental():
    """finalize, ingest more, finalize again (no reset): the second table is that of all reads; finalize twice in a row
    gives the same table; table growth / rehash happens in between (tiny capacity hint)"""
    reads = rand_reads(91, 120, 2000, 9000)
    reads += [r[30:] for r in reads[:80]]
    k, l, d, A = 5, 12, 0.006, 2
    R = _mdbg()
    with R.Mdbg(k, l, d, A, table_capacity_hint=16) as m:
        m.ingest_reads(reads[:60], 0)
        first = m.finalize()
        assert_nodes_equal(first, oracle_graph(reads[:60], k, l, d, A))
        assert_nodes_equal(m.finalize(), first)
        m.ingest_reads(reads[60:150], 60)
        m.ingest_reads(reads[150:], 150)
        got = m.finalize()
        e1 = m.graph_edges(0.01)
        assert_nodes_equal(m.finalize(), got)
        e2 = m.graph_edges(0.01)
    assert_nodes_equal(got, oracle_graph(reads, k, l, d, A))
    assert all(np.array_equal(e1[f], e2[f]) for f in ("n1", "o1", "n2", "o2", "overlap"))


@pytest.mark.parametrize("k", [300, 1000, 4096])
def test_large_k(k):
    """large k: dynamic LDS of the insert kernels (256 + k keys), the overlapping 8-value compare rounds, edges over long keys"""
    from rust_mdbg_amd import emit as E
    base = rand_reads(500 + k, 6, 60000, 90000)
    reads = base + [r[200:] for r in base] + [base[0][:30000]]
    l, d, A = 10, 0.05, 2
    exp = oracle_graph(reads, k, l, d, A)
    R = _mdbg()
    with R.Mdbg(k, l, d, A) as m:
        m.ingest_reads(reads, 0)
        got = m.finalize()
        ge = m.graph_edges(0.01)
    assert_nodes_equal(got, exp)
    assert exp["n_nodes"] > 100
    he = E.Emitter().edges(got, 0.01)
    assert all(np.array_equal(ge[f], he[f]) for f in ("n1", "o1", "n2", "o2", "overlap")) and len(ge["n1"]) > 100


@pytest.mark.parametrize("d", [0.0, 1e-30, 0.999999, 1.0, 1.5, 7.0])
def test_density_extremes(d):
    """hash_bound = floor(d * 2^64) saturates like the reference's `as u64` cast (src/read.rs:183): d = 0 selects (almost) nothing,
    d >= 1 selects every l-mer"""
    reads = rand_reads(3, 12, 500, 3000)
    k, l, A = 3, 9, 1
    b, o = O.concat_reads(reads)
    exp_sk = O.sketch(b, o, l, d)
    R = _mdbg()
    with R.Mdbg(k, l, d, A) as m:
        assert_sketch_equal(m.sketch(b, o), exp_sk)
        m.ingest(b, o, 0)
        got = m.finalize()
    assert_nodes_equal(got, oracle_graph(reads, k, l, d, A))
    assert (int(exp_sk["off"][-1]) == 0) == (d < 1e-20)


def test_large_and_sparse_read_ordinals():
    """read ordinals are caller-defined (position of the record in the input): huge values and gaps between batches are fine,
    the order of first sightings follows the ordinals, not the call order"""
    reads = rand_reads(17, 60, 2000, 6000)
    reads += [r[10:] for r in reads[:40]]
    k, l, d, A = 4, 12, 0.006, 2
    cuts = [(0, 30, (1 << 37) + 5), (30, 70, 12), (70, 100, (1 << 37) + 1000)]          # (lo, hi, first ordinal): the middle batch comes FIRST in ordinal order
    g = O.Graph(k, l, d, A)
    for lo, hi, first in sorted(cuts, key=lambda c: c[2]):
        b, o = O.concat_reads(reads[lo:hi])
        assert g.ingest(b, o, first) == 0
    exp = g.finalize(with_edges=False)
    R = _mdbg()
    with R.Mdbg(k, l, d, A) as m:
        for lo, hi, first in cuts:
            m.ingest_reads(reads[lo:hi], first)
        got = m.finalize()
        with pytest.raises(R.MdbgError) as ei:
            m.ingest_reads(reads[:2], 1 << 38)                      # ordinal << 26 must fit in 64 bits
        assert ei.value.code == -3
    assert_nodes_equal(got, exp)
    assert int(got["src_read"].max()) > (1 << 37) and exp["n_nodes"] > 500


@pytest.mark.parametrize("k", [2047, 4096])
def test_very_long_k(k):
    """k up to 4096: the insert kernels stage 2048 + k - 1 hashes per workgroup (48 KB of LDS at k = 4096)"""
    from rust_mdbg_amd import synth
    rs = synth.synth_reads(3, 60000, 2, mean_len=30000, sd_len=2000, min_len=26000, max_len=34000, err_ppm=0)
    reads = [rs[0], rs[1], rs[0], O.revcomp(rs[1])]
    bases, offs = O.concat_reads(reads)
    g = O.Graph(k, 8, 0.5, 2)
    g.ingest(bases, offs)
    exp = g.finalize(with_edges=False)
    R = _mdbg()
    with R.Mdbg(k, 8, 0.5, 2) as m:
        m.ingest(bases, offs, 0)
        got = m.finalize()
    assert exp["n_nodes"] > 1000
    assert_nodes_equal(got, exp)


def test_overlapping_read_ordinals_are_an_error():
    """two batches whose ordinal ranges overlap (a caller that always passes 0) must not yield a silently wrong table"""
    R = _mdbg()
    reads = rand_reads(7, 40, 3000, 6000)
    with R.Mdbg(5, 10, 0.02, 2) as m:
        m.ingest_reads(reads[:20], 0)
        m.ingest_reads(reads[20:], 10)                               # [10, 30) overlaps [0, 20)
        with pytest.raises(R.MdbgError) as ei:
            m.finalize()
        assert ei.value.code == -1 and "overlap" in str(ei.value)
        m.reset(0)                                                   # the context stays usable
        m.ingest_reads(reads[:20], 0)
        m.ingest_reads(reads[20:], 20)
        assert m.finalize()["n_nodes"] >= 0


def test_param_validation():
    R = _mdbg()
    for kw in (dict(k=1, l=12, density=0.01), dict(k=5, l=1, density=0.01), dict(k=5, l=256, density=0.01),
               dict(k=5, l=12, density=0.01, min_abundance=0), dict(k=5, l=12, density=0.01, min_abundance=65536)):
        with pytest.raises(R.MdbgError) as e:
            R.Mdbg(**kw)
        assert e.value.code == -1


def test_device_synth_matches_cpu_regenerator():
    from rust_mdbg_amd import synth
    R = _mdbg()
    kw = dict(mean_len=6000, sd_len=1500, min_len=500, max_len=12000, err_ppm=3000)
    with R.Mdbg(5, 12, 0.01, 2) as m:
        db, do, nb = m.synth_reads_device(seed=9, genome_len=123457, n_reads=64, first_read=1000, **kw)
        cpu = synth.synth_reads(9, 123457, 64, first_read=1000, **kw)
        assert nb == sum(map(len, cpu))
        hb = m.to_host(db, nb)
        ho = m.to_host(do, 65 * 8, np.uint64)
        assert hb.tobytes() == b"".join(cpu)
        assert ho.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in cpu])]).tolist()
        # and the device-resident ingest path equals the host path on the same bytes
        m.ingest_device(db, do, 64, nb, 0)
        got = m.finalize()
    exp = oracle_graph(cpu, 5, 12, 0.01, 2)
    assert_nodes_equal(got, exp)


@pytest.mark.parametrize("A", [9, 12, 40, 300])
def test_large_min_abundance(A):
    """minabund above what the table slots track (8): the A-th sighting of every solid node comes from the re-scan at finalize
    (the reference's DbgAbundance is a u16: any value up to 65535 is legal, src/main.rs:60,680)"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(31 + A, 3000, 6000 if A >= 40 else 900, mean_len=2500, sd_len=300, min_len=800, max_len=4000, err_ppm=3000)
    k, l, d = 4, 8, 0.02
    exp = oracle_graph(reads, k, l, d, A)
    assert exp["n_nodes"] > 20 and exp["n_nodes_before"] > exp["n_nodes"]
    got, _ = run_gpu(reads, k, l, d, A)
    assert_nodes_equal(got, exp)
    got2, _ = run_gpu(reads, k, l, d, A, batches=[(0, len(reads) // 3), (len(reads) // 3, len(reads))])
    assert_nodes_equal(got2, exp)


@pytest.mark.parametrize("l", [33, 40, 64, 100])
def test_l_above_32_runs_on_the_generic_walker(l):
    reads = rand_reads(l, 12, 50, 30000, hp=0.05)
    b, o = O.concat_reads(reads)
    exp = O.sketch(b, o, l, 0.01)
    R = _mdbg()
    with R.Mdbg(5, l, 0.01, 2) as m:
        assert_sketch_equal(m.sketch(b, o), exp)
        m.ingest(b, o, 0)
        got = m.finalize()
        st = m.stats()
    assert st["n_slow_tiles"] == st["n_tiles"] > 0
    assert_nodes_equal(got, oracle_graph(reads, 5, l, 0.01, 2))


@pytest.mark.parametrize("k,l,d,A", [(5, 10, 0.01, 2), (9, 12, 0.004, 1), (3, 8, 0.05, 3), (4, 8, 0.02, 12)])
def test_read_stats_query(k, l, d, A):
    """--read_stats (src/main.rs:939-1004): abundance of every k-min-mer of a query read in the FILTERED table, 0 when absent"""
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(5 + k, 150000, 260, mean_len=9000, sd_len=1500, min_len=2000, max_len=16000, err_ppm=3000)
    query = synth.synth_reads(5 + k, 150000, 60, mean_len=9000, sd_len=1500, min_len=2000, max_len=16000, err_ppm=20000, first_read=100)
    query += [b"", b"ACGT" * 3, rand_reads(1, 1, 30000, 30000)[0], reads[0]]
    exp_nodes = oracle_graph(reads, k, l, d, A)
    table = {tuple(int(x) for x in exp_nodes["keys"][i]): int(exp_nodes["abundance"][i]) for i in range(exp_nodes["n_nodes"])}
    qb, qo = O.concat_reads(query)
    sk = O.sketch(qb, qo, l, d)
    exp_counts, exp_off = [], [0]
    for r in range(len(query)):
        h = [int(x) for x in sk["hashes"][int(sk["off"][r]):int(sk["off"][r + 1])]]
        if len(h) > k:
            for i in range(len(h) - k + 1):
                w = tuple(h[i:i + k]); rv = w[::-1]
                exp_counts.append(table.get(w if w < rv else rv, 0))
        exp_off.append(len(exp_counts))
    R = _mdbg()
    with R.Mdbg(k, l, d, A) as m:
        m.ingest_reads(reads, 0)
        before = m.finalize()
        counts, off = m.query(qb, qo)
        st = m.stats()
        after = m.finalize()
    assert off.tolist() == exp_off and counts.tolist() == exp_counts
    assert sum(1 for c in exp_counts if c) > 50 and sum(1 for c in exp_counts if c == 0) > 50
    assert st["n_reads"] == len(reads)                       # the query left the resident sketches and the table alone
    assert_nodes_equal(after, exp_nodes) and assert_nodes_equal(before, exp_nodes) is None
