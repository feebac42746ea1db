"""Host packer of the 2-bit input layout (libmdbg_emit.so: mdbg_pack_reads), CPU only."""
import numpy as np
import pytest

from oracle import oracle as O


def unpack(words, n):
    lo = (words & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi = (words >> np.uint64(32)).astype(np.uint64)
    i = np.arange(n, dtype=np.uint64)
    w, b = (i >> np.uint64(5)).astype(np.int64), i & np.uint64(31)
    return (((lo[w] >> b) & np.uint64(1)) | (((hi[w] >> b) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 1000, 300001])
def test_pack_reads_matches_the_layout(n, threads):
    from rust_mdbg_amd import emit as E
    rng = np.random.default_rng(n + threads)
    b = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).astype(np.uint8)
    exc = {}
    for p in rng.choice(n, size=min(n, 17), replace=False) if n else []:
        b[p] = rng.choice(np.frombuffer(b"NnacgtX-\n", dtype=np.uint8)); exc[int(p)] = int(b[p])
    pk = E.pack_reads(b, np.array([0, n], dtype=np.uint64), threads=threads)
    assert len(pk["words"]) == (n + 31) // 32
    assert pk["exc_pos"].tolist() == sorted(exc) and pk["exc_val"].tolist() == [exc[p] for p in sorted(exc)]
    codes = unpack(pk["words"], n)
    ok = np.ones(n, bool); ok[list(exc)] = False
    assert np.array_equal(codes[ok], ((b >> 1) & 3)[ok])
    if n % 32:                                   # bits past the last base are zero
        assert int(pk["words"][-1]) >> (32 + n % 32) == 0 and (int(pk["words"][-1]) & 0xFFFFFFFF) >> (n % 32) == 0


def test_pack_reads_small_exception_capacity_reports_the_count():
    from rust_mdbg_amd import emit as E
    b = np.frombuffer(b"ACGTNNNNACGT" * 10, dtype=np.uint8)
    with pytest.raises(RuntimeError):
        E.pack_reads(b, np.array([0, len(b)], dtype=np.uint64), exc_cap=3)
    pk = E.pack_reads(b, np.array([0, len(b)], dtype=np.uint64))
    assert len(pk["exc_pos"]) == 40 and set(pk["exc_val"].tolist()) == {ord("N")}
