"""The oracle against the committed config-1 fixtures (example/reads-0.00.fa.gz, k=7 l=10 d=0.0008 A=2)."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN
from oracle import oracle as O
from golden.make_golden import edge_sha, node_sha


def test_example_cfg1(example_reads):
    gold = json.load(open(os.path.join(GOLDEN, "example_cfg1.json")))
    c = gold["config"]
    assert len(example_reads) == gold["n_reads"] == 657
    bases, offs = O.concat_reads(example_reads)
    assert int(offs[-1]) == gold["n_bases"] == 14744805
    assert O.hash_bound(c["density"]) == gold["hash_bound"]
    sk = O.sketch(bases, offs, c["l"], c["density"])
    assert sk["err"] == 0 and len(sk["hashes"]) == gold["n_minimizers"] == 16069
    o = sk["off"]
    assert int(o[1] - o[0]) == gold["read0_n"]
    assert [[int(sk["pos"][i]), int(sk["hashes"][i])] for i in range(3)] == gold["read0_first3"]
    assert hashlib.sha256(sk["hashes"].tobytes() + sk["pos"].tobytes() + sk["off"].tobytes()).hexdigest() == gold["minimizers_sha256"]
    g = O.Graph(c["k"], c["l"], c["density"], c["minabund"])
    assert g.ingest(bases, offs) == 0
    r = g.finalize()
    for f in ("n_windows", "n_nodes_before", "n_nodes", "n_edges", "presimp_removed"):
        assert r[f] == gold[f], f
    # the independent restatement (tests/golden/independent_restatement.py, committed; tests/test_oracle_independent.py) regenerates this very digest
    assert node_sha(r["keys"], r["abundance"]) == gold["nodes_sha256"] == "89e36af94df5e2227ded239b9d3423eb654131f2d58721ca7b873b2ef86feab8"
    assert edge_sha(r) == gold["edges_sha256"]
    z = np.load(os.path.join(GOLDEN, "example_cfg1_nodes.npz"))
    for f in ("keys", "index", "abundance", "seqlen", "shift", "src_read", "src_start", "src_end", "reversed", "edge_n1", "edge_overlap"):
        assert np.array_equal(z[f], r[f]), f
    # threaded timing variant agrees on the counts
    solid, wins = O.count_threaded(bases, offs, c["k"], c["l"], c["density"], c["minabund"], threads=4)
    assert (solid, wins) == (gold["n_nodes"], gold["n_windows"])


def test_oracle_against_the_references_own_python_helpers():
    """tests/golden/reference_py_vectors.json holds what two of the reference's Python utilities print when run on committed inputs
    (tests/golden/make_reference_py_vectors.py ran them from /root/reference in the build container): utils/remove_homopoly.py — the homopolymer
    compression of src/read.rs:157-174, same literal "ACTGactgNn" — and utils/parse_gfa.py — how the reference's tools read S lines.
    The oracle's encode_rle must give the compressed string of every case, with positions that are the run starts."""
    v = json.load(open(os.path.join(GOLDEN, "reference_py_vectors.json")))
    assert len(v["hpc"]) == 80
    changed = 0
    for case in v["hpc"]:
        s = case["input"].encode()
        hpc, pos = O.encode_rle(s)
        assert hpc.decode() == case["output"], case
        assert len(pos) == len(hpc) and all(s[p] == hpc[i] for i, p in enumerate(pos)) and pos == sorted(set(pos))
        assert all(p == 0 or not (s[p - 1] == s[p] and s[p:p + 1] in b"ACTGactgNn") for p in pos)      # every kept position starts a run
        changed += hpc != s
    assert changed > 40


def test_emitted_gfa_is_read_by_the_references_parser_as_meant(example_reads, tmp_path):
    """the S lines this framework's emitter writes for BASELINE configs[0] are the ones the reference's utils/parse_gfa.py was run on when the vectors were
    made, and what that parser read out of them (id -> KC abundance) is the oracle's node table"""
    from rust_mdbg_amd.emit import Emitter
    v = json.load(open(os.path.join(GOLDEN, "reference_py_vectors.json")))
    b, o = O.concat_reads(example_reads)
    g = O.Graph(7, 10, 0.0008, 2)
    g.ingest(b, o)
    r = g.finalize(with_edges=True)
    em = Emitter()
    em.edges(r, 0.01)
    p = str(tmp_path / "cfg1.gfa")
    em.write_gfa(p, r)
    text = open(p).read()
    assert hashlib.sha256(text.encode()).hexdigest() == v["gfa_text_sha256"]
    assert [ln for ln in text.split("\n") if ln.startswith("S")] == v["gfa_s_lines"]
    assert v["gfa_abundance"] == {str(int(r["index"][i])): int(r["abundance"][i]) for i in range(r["n_nodes"])} and len(v["gfa_abundance"]) == 104
