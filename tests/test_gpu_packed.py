"""The 2-bit packed input path (mdbg_ingest_batch_packed / _device, mdbg_pack_device) against the oracle and against the ASCII path."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import assert_nodes_equal, oracle_graph, rand_reads

pytestmark = pytest.mark.gpu
FIELDS = ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed")


def packed_sketch(reads, l, d, hpc=False, k=5):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    b, o = O.concat_reads(reads)
    with R.Mdbg(k, l, d, 2, reads_already_hpc=hpc) as m:
        m.ingest_packed(E.pack_reads(b, o, threads=2), 0)
        sk = m.store_sketch()
        st = m.stats()
    return sk, st, O.sketch(b, o, l, d, hpc)


@pytest.mark.parametrize("l,d", [(10, 0.0008), (12, 0.003), (14, 0.003), (12, 0.002), (5, 0.01), (20, 0.01), (31, 0.02), (12, 0.1)])
@pytest.mark.parametrize("hpc", [False, True])
def test_packed_sketch_random_reads(l, d, hpc):
    reads = rand_reads(300 + l, 40, 0, 60000, hp=0.05)
    reads += [b"", b"A", b"ACGT" * 2, b"", b"C" * 500, rand_reads(5, 1, 200000, 200000)[0], b""]
    sk, st, exp = packed_sketch(reads, l, d, hpc)
    # (the 500-base homopolymer read may sit in front of a tile boundary; wave tiles — MDBG_TILE=1x4 — look back 128 bases and have four times the boundaries)
    assert exp["err"] == 0 and st["n_slow_tiles"] <= (1 if st["tile_bases"] > 20000 else 16)
    for f in ("hashes", "pos", "off"):
        assert np.array_equal(sk[f], exp[f]), f


def test_packed_with_n_exceptions_and_homopolymers_across_words():
    rng = np.random.default_rng(5)
    reads = []
    for i in range(30):
        x = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(100, 90000))).tobytes())
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, len(x))); x[p:p + int(rng.integers(1, 40))] = b"N" * len(x[p:p + int(rng.integers(1, 40))])
        for _ in range(int(rng.integers(0, 6))):                 # homopolymers that straddle 32-base words and tiles
            p = int(rng.integers(0, len(x))); n = int(rng.choice([2, 31, 32, 33, 64, 65, 300, 40000])); x[p:p + n] = bytes([x[p]]) * len(x[p:p + n])
        reads.append(bytes(x))
    sk, st, exp = packed_sketch(reads, 12, 0.01)
    assert exp["err"] == 0 and st["n_slow_tiles"] > 0
    for f in ("hashes", "pos", "off"):
        assert np.array_equal(sk[f], exp[f]), f


@pytest.mark.parametrize("bad", [b"a", b"-", b"\n", b"X"])
def test_packed_alphabet_error_rule(bad):
    """a byte outside ACGTN raises MDBG_E_ALPHABET iff its read has at least l HPC bases (the reference's nthash panic), also through the side-list"""
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    rng = np.random.default_rng(9)
    long_read = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=5000).tobytes())
    short = b"ACG" + bad + b"AC"                                # 6 HPC bases < l: tolerated, no minimizer
    b, o = O.concat_reads([bytes(long_read), short])
    with R.Mdbg(5, 12, 0.01, 2) as m:
        m.ingest_packed(E.pack_reads(b, o), 0)
        sk = m.store_sketch()
    exp = O.sketch(b, o, 12, 0.01)
    assert exp["err"] == 0 and np.array_equal(sk["hashes"], exp["hashes"])
    long_read[2500] = bad[0]
    b, o = O.concat_reads([bytes(long_read), short])
    assert O.sketch(b, o, 12, 0.01)["err"] != 0
    with R.Mdbg(5, 12, 0.01, 2) as m:
        with pytest.raises(R.MdbgError) as ei:
            m.ingest_packed(E.pack_reads(b, o), 0)
        assert ei.value.code == -2


def test_packed_graph_equals_oracle_and_ascii_path():
    from rust_mdbg_amd import synth
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    reads = synth.synth_reads(21, 400000, 500, mean_len=12000, sd_len=1500, min_len=3000, max_len=20000, err_ppm=2000)
    k, l, d, a = 15, 12, 0.004, 2
    exp = oracle_graph(reads, k, l, d, a)
    b, o = O.concat_reads(reads)
    half = len(reads) // 2
    cut = int(o[half])
    with R.Mdbg(k, l, d, a) as m:                                # two packed batches, second one first
        m.ingest_packed(E.pack_reads(b[cut:], o[half:] - o[half]), half)
        m.ingest_packed(E.pack_reads(b[:cut], o[:half + 1]), 0)
        got = m.finalize()
    assert_nodes_equal(got, exp)


def test_pack_device_equals_host_packer_and_feeds_the_kernel():
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    k, l, d, a = 21, 12, 0.003, 2
    n_reads = 3000
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=3, genome_len=5_000_000, n_reads=n_reads)
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        ep = torch.zeros(16, dtype=torch.int64, device="cuda"); ev = torch.zeros(16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
        assert m.pack_device(db, nb, words.data_ptr(), ep.data_ptr(), ev.data_ptr(), 16) == 0
        bases = m.to_host(db, nb); offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
        pk = E.pack_reads(bases, offs, threads=4)
        assert np.array_equal(words.cpu().numpy().view(np.uint64)[:len(pk["words"])], pk["words"])
        m.ingest_device(db, do, n_reads, nb, 0)
        one = m.finalize(); st1 = m.stats()
        m.reset(0)
        m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0)
        two = m.finalize(); st2 = m.stats()
    assert one["n_nodes"] == two["n_nodes"] > 1000 and st1["n_minimizers"] == st2["n_minimizers"]
    for f in FIELDS:
        assert np.array_equal(one[f], two[f]), f
