"""Host reader (libmdbg_emit.so: mdbg_reader_*): FASTA/FASTQ, gzip, batching, seq_io-style multi-line records."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN
from rust_mdbg_amd import emit as E


def collect(path, max_bases=1 << 30, strip=False):
    out = []
    with E.Reader(path, strip) as r:
        fasta = r.is_fasta
        for b, o in r.batches(max_bases):
            assert o[0] == 0 and len(b) == o[-1]
            out += [b[int(o[i]):int(o[i + 1])].tobytes() for i in range(len(o) - 1)]
    return out, fasta


def test_example_fasta_gz(example_reads):
    reads, fasta = collect(os.path.join(GOLDEN, "reads-0.00.fa.gz"))
    assert fasta and reads == example_reads
    # small batches: whole records, every batch within the limit unless a single record exceeds it
    n = 0
    with E.Reader(os.path.join(GOLDEN, "reads-0.00.fa.gz")) as r:
        for b, o in r.batches(50000):
            assert len(o) - 1 >= 1 and (len(b) <= 50000 or len(o) == 2)
            n += len(o) - 1
    assert n == 657


def test_formats(tmp_path):
    fq = tmp_path / "r.fastq"
    fq.write_bytes(b"@r1 x\nACGTAC\n+\nIIIIII\n@r2\nGGGTTT\r\n+\r\nIIIIII\r\n@r3\nAC\n+\nII")
    assert collect(str(fq)) == ([b"ACGTAC", b"GGGTTT", b"AC"], False)
    fqz = tmp_path / "r.fq.gz"
    with gzip.open(fqz, "wb") as f:
        f.write(fq.read_bytes())
    assert collect(str(fqz)) == ([b"ACGTAC", b"GGGTTT", b"AC"], False)          # ".fq.gz" is FASTQ (name rule of main.rs:463)
    fa = tmp_path / "m.fasta"
    fa.write_bytes(b">a\nACGT\nTTGA\n>b desc\nCC\r\nGG\r\n>c\nA")
    assert collect(str(fa)) == ([b"ACGT\nTTGA", b"CC\r\nGG", b"A"], True)       # seq_io keeps interior terminators
    assert collect(str(fa), strip=True) == ([b"ACGTTTGA", b"CCGG", b"A"], True)  # --reference behaviour
    odd = tmp_path / "x.fa.txt"
    odd.write_bytes(b">a\nAC\n")
    assert collect(str(odd))[1] is True                                          # ".fa." inside the name
    with pytest.raises(OSError):
        E.Reader(str(tmp_path / "missing.fa"))
    with pytest.raises(OSError):
        E.Reader(str(tmp_path / "reads.fa.lz4"))
    for s in E.READER_EXPORTS:
        assert hasattr(E.load_library(), s)
