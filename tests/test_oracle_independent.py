"""The C++ oracle against the SECOND restatement of the reference (tests/golden/independent_restatement.py: Python/numpy, direct
ntHash definition, no code shared with oracle/): committed fixtures of the reference's example file (BASELINE configs[0]) and of
48 seeded small cases with complete expected outputs (sketches, node tables field by field, edge multisets, the alphabet error).
The reference itself cannot be run here (Rust, no cargo), so two independently written restatements agreeing is the pin."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

CASES = json.load(open(os.path.join(GOLDEN, "independent_cases.json")))["cases"]


def test_fixtures_are_what_the_restatement_generates_today():
    from golden import independent_restatement as I
    assert I.config1() == json.load(open(os.path.join(GOLDEN, "independent_cfg1.json")))
    fresh = json.loads(json.dumps(I.random_cases()))
    assert fresh == CASES


def test_oracle_equals_independent_restatement_on_the_example_file(example_reads):
    gold = json.load(open(os.path.join(GOLDEN, "independent_cfg1.json")))
    c = gold["config"]
    bases, offs = O.concat_reads(example_reads)
    assert O.hash_bound(c["density"]) == gold["hash_bound"]
    sk = O.sketch(bases, offs, c["l"], c["density"])
    import hashlib
    assert hashlib.sha256(sk["hashes"].tobytes() + sk["pos"].tobytes() + sk["off"].tobytes()).hexdigest() == gold["minimizers_sha256"]
    g = O.Graph(c["k"], c["l"], c["density"], c["minabund"])
    assert g.ingest(bases, offs) == 0
    r = g.finalize()
    for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "n_edges", "presimp_removed"):
        assert r[f] == gold[f], f
    from golden.make_golden import edge_sha, node_sha
    assert node_sha(r["keys"], r["abundance"]) == gold["nodes_sha256"] and edge_sha(r) == gold["edges_sha256"]


def oracle_case(c):
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    g = O.Graph(c["k"], c["l"], c["density"], c["minabund"], c["already_hpc"], c["presimp"])
    err = g.ingest(bases, offs)
    return bases, offs, g, err


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_oracle_equals_independent_restatement_case(ci):
    c = CASES[ci]
    bases, offs, g, err = oracle_case(c)
    sk = O.sketch(bases, offs, c["l"], c["density"], c["already_hpc"])
    if "error_read" in c:
        assert err != 0 and sk["err"] != 0
        return
    assert err == 0 and sk["err"] == 0
    o = sk["off"]
    for i, (pos, hs) in enumerate(c["sketch"]):
        assert sk["pos"][int(o[i]):int(o[i + 1])].tolist() == pos and sk["hashes"][int(o[i]):int(o[i + 1])].tolist() == hs, ("sketch of read", i)
    r = g.finalize()
    for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed"):
        assert r[f] == c[f], f
    for row, n in enumerate(c["nodes"]):
        assert r["keys"][row].tolist() == n["key"] and int(r["index"][row]) == n["index"] and int(r["abundance"][row]) == n["abundance"]
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and r["shift_full"][row].tolist() == n["shift_full"]
        assert int(r["reversed"][row]) == n["reversed"] and int(r["src_read"][row]) == n["src_read"]
        assert int(r["src_start"][row]) == n["src_start"] and int(r["src_end"][row]) == n["src_end"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in
                 zip(r["edge_n1"], r["edge_o1"], r["edge_n2"], r["edge_o2"], r["edge_overlap"]))
    assert got == sorted(c["edges"])


SYNC_CASES = json.load(open(os.path.join(GOLDEN, "independent_syncmer_cases.json")))["cases"]


def test_syncmer_fixtures_are_what_the_restatement_generates_today():
    from golden import independent_restatement as I
    assert json.loads(json.dumps(I.syncmer_cases())) == SYNC_CASES


@pytest.mark.parametrize("ci", range(len(SYNC_CASES)))
def test_oracle_syncmers_equal_independent_restatement(ci):
    c = SYNC_CASES[ci]
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    sk = O.sketch(bases, offs, c["l"], c["density"], c["already_hpc"], syncmer_s=c["syncmer_s"])
    assert sk["err"] == 0
    o = sk["off"]
    for i, (pos, hs) in enumerate(c["sketch"]):
        assert sk["pos"][int(o[i]):int(o[i + 1])].tolist() == pos and sk["hashes"][int(o[i]):int(o[i + 1])].tolist() == hs, ("sketch of read", i)
    g = O.Graph(c["k"], c["l"], c["density"], c["minabund"], c["already_hpc"], c["presimp"], syncmer_s=c["syncmer_s"])
    assert g.ingest(bases, offs) == 0
    r = g.finalize()
    for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed"):
        assert r[f] == c[f], f
    for row, n in enumerate(c["nodes"]):
        assert r["keys"][row].tolist() == n["key"] and int(r["index"][row]) == n["index"] and int(r["abundance"][row]) == n["abundance"]
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and int(r["src_start"][row]) == n["src_start"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in
                 zip(r["edge_n1"], r["edge_o1"], r["edge_n2"], r["edge_o2"], r["edge_overlap"]))
    assert got == sorted(c["edges"])


LMER_CASES = json.load(open(os.path.join(GOLDEN, "independent_lmer_cases.json")))["cases"]


def test_lmer_fixtures_are_what_the_restatement_generates_today():
    from golden import independent_restatement as I
    assert json.loads(json.dumps(I.lmer_cases())) == LMER_CASES


@pytest.mark.parametrize("ci", range(len(LMER_CASES)))
def test_oracle_lmer_counts_equal_independent_restatement(ci):
    """--lmer-counts (src/main.rs:544-566, src/minimizers.rs:53-113, src/read.rs:200-205): selection of the l-mers, filtered sketch, graph"""
    c = LMER_CASES[ci]
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    m = O.LmerMap([(w.encode(), n) for w, n in c["lmer_lines"]], c["l"], c["density"], c["lmer_min"], c["lmer_max"])
    assert m.err == 0
    assert [[w.decode(), h] for w, h in m.selected()] == c["selected"]
    sk = O.sketch(bases, offs, c["l"], c["density"], c["already_hpc"], lmer_map=m)
    assert sk["err"] == 0
    o = sk["off"]
    for i, (pos, hs) in enumerate(c["sketch"]):
        assert sk["pos"][int(o[i]):int(o[i + 1])].tolist() == pos and sk["hashes"][int(o[i]):int(o[i + 1])].tolist() == hs, ("sketch of read", i)
    g = O.Graph(c["k"], c["l"], c["density"], c["minabund"], c["already_hpc"], c["presimp"], lmer_map=m)
    assert g.ingest(bases, offs) == 0
    r = g.finalize()
    for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed"):
        assert r[f] == c[f], f
    for row, n in enumerate(c["nodes"]):
        assert r["keys"][row].tolist() == n["key"] and int(r["index"][row]) == n["index"] and int(r["abundance"][row]) == n["abundance"]
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and int(r["src_start"][row]) == n["src_start"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in
                 zip(r["edge_n1"], r["edge_o1"], r["edge_n2"], r["edge_o2"], r["edge_overlap"]))
    assert got == sorted(c["edges"])
