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
