"""--lmer-counts, host side: libmdbg_emit's counts-file reader and selection rule (src/main.rs:544-566, src/minimizers.rs:53-113) against
the independent restatement's fixtures and the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from rust_mdbg_amd import emit as E

CASES = json.load(open(os.path.join(GOLDEN, "independent_lmer_cases.json")))["cases"]


def decode(code, l):
    return "".join("ACTG"[(int(code) >> (2 * (l - 1 - i))) & 3] for i in range(l))


def write_counts(path, lines):
    with open(path, "w") as f:
        for i, (w, c) in enumerate(lines):
            f.write("%s%s%d\n" % (w, "\t" if i % 2 else " ", c))       # kmc_dump writes tabs; split_whitespace takes both


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_counts_file_selection_matches_fixture(ci, tmp_path):
    c = CASES[ci]
    p = str(tmp_path / "counts.txt")
    write_counts(p, c["lmer_lines"])
    codes, ignored = E.lmer_filter_from_counts(p, c["l"], c["density"], c["lmer_min"], c["lmer_max"])
    assert ignored == 0
    assert codes.tolist() == sorted(set(codes.tolist()))
    assert sorted(decode(x, c["l"]) for x in codes) == [w for w, _ in c["selected"]]


def test_counts_file_edge_cases(tmp_path):
    l, d = 6, 0.995          # (at density 1.0 the reference's "skip" has no effect: it sets the ratio to 1.0 and tests ratio <= density)
    p = str(tmp_path / "c.txt")
    # wrong length and non-ACGT l-mers are ignored; a later line for the same canonical l-mer wins; thresholds are exclusive
    open(p, "w").write("ACGTAC 5\nACGTACG 5\nACGTNC 5\nGTACGT 1\nTTTTTT 9\nCCCCCC 2\nGGGGGA 3\n  AAGCTT\t7  \n")
    codes, ignored = E.lmer_filter_from_counts(p, l, d, 2, 9)
    got = sorted(decode(x, l) for x in codes)
    # ACGTAC/GTACGT are each other's reverse complement: the later line (count 1 <= min) wins -> skipped; TTTTTT = AAAAAA count 9 >= max -> skipped;
    # CCCCCC count 2 <= min -> skipped; GGGGGA (canonical TCCCCC) count 3 kept with its reverse complement; AAGCTT is its own reverse complement
    assert set(got) <= {"GGGGGA", "TCCCCC", "AAGCTT"} and len(got) >= 2 and ignored == 2
    m = O.LmerMap([(b"ACGTAC", 5), (b"GTACGT", 1), (b"TTTTTT", 9), (b"CCCCCC", 2), (b"GGGGGA", 3), (b"AAGCTT", 7)], l, d, 2, 9)
    assert sorted(w.decode() for w, _ in m.selected()) == got
    # the density rule in f64: only l-mers whose hash / 2^64 <= d
    codes2, _ = E.lmer_filter_from_counts(p, l, 0.3, 2, 9)
    m2 = O.LmerMap([(b"GGGGGA", 3), (b"AAGCTT", 7)], l, 0.3, 2, 9)
    assert sorted(decode(x, l) for x in codes2) == sorted(w.decode() for w, _ in m2.selected())
    for bad in ("ACGTAC\n", "ACGTAC x\n", "ACGTAC -1\n", "ACGTAC 4294967296\n"):
        open(p, "w").write(bad)
        with pytest.raises(RuntimeError, match="-1"):
            E.lmer_filter_from_counts(p, l, d)
    with pytest.raises(RuntimeError, match="-7"):
        E.lmer_filter_from_counts(str(tmp_path / "missing.txt"), l, d)
    open(p, "w").write("")
    codes3, _ = E.lmer_filter_from_counts(p, l, d)
    assert len(codes3) == 0
    # density 1.0: skipped l-mers come back (ratio 1.0 <= 1.0), exactly as in the reference
    open(p, "w").write("CCCCCC 2\n")
    codes4, _ = E.lmer_filter_from_counts(p, l, 1.0, 2, 9)
    assert sorted(decode(x, l) for x in codes4) == ["CCCCCC", "GGGGGG"] == sorted(w.decode() for w, _ in O.LmerMap([(b"CCCCCC", 2)], l, 1.0, 2, 9).selected())
