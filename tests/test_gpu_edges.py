"""GPU edge construction (mdbg_graph_edges, csrc/edges.hip) == the host emitter == the oracle's end-to-end edges
(src/main.rs:1017-1117), in the same order, with and without presimplification."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_fuzz import fuzz_reads
from test_gpu_parity import _mdbg, assert_nodes_equal

pytestmark = pytest.mark.gpu

F = ("n1", "o1", "n2", "o2", "overlap")


def as_rows(e):
    return list(zip(*(np.asarray(e[f]).tolist() for f in F)))


@pytest.mark.parametrize("presimp", [0.0, 0.01, 0.5])
@pytest.mark.parametrize("seed", range(6))
def test_gpu_edges_equal_host_emitter_and_oracle(seed, presimp):
    from rust_mdbg_amd import emit as E
    R = _mdbg()
    rnd = random.Random(500 + seed)
    k, l, d, A = rnd.choice([(2, 8, 0.03, 1), (3, 8, 0.03, 1), (5, 10, 0.01, 2), (7, 12, 0.008, 2), (4, 6, 0.05, 3), (12, 12, 0.01, 1)])
    reads = fuzz_reads(rnd, n_reads=200, genome_len=rnd.choice([3000, 30000]), mean_len=4000, err=rnd.choice([0.0, 0.01]), p_lower=0.0, p_n=0.0,
                       p_hp=rnd.choice([0.0, 0.02]))
    bases, offs = O.concat_reads(reads)
    with R.Mdbg(k, l, d, A) as m:
        m.ingest(bases, offs, 0)
        nodes = m.finalize()
        got = m.graph_edges(presimp)
        again = m.graph_edges(presimp)                           # buffers are reused: same answer
    host = E.Emitter().edges(nodes, presimp)
    assert as_rows(got) == as_rows(host) == as_rows(again)       # identical ORDER, not only the same multiset
    assert got["presimp_removed"] == host["presimp_removed"]
    g = O.Graph(k, l, d, A, presimp=presimp)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=True)
    assert_nodes_equal(nodes, exp)
    exp_rows = sorted(zip(exp["edge_n1"].tolist(), exp["edge_o1"].tolist(), exp["edge_n2"].tolist(), exp["edge_o2"].tolist(), exp["edge_overlap"].tolist()))
    assert sorted(as_rows(got)) == exp_rows and got["presimp_removed"] == exp["presimp_removed"]
    if seed == 0 and presimp == 0.01:
        assert len(exp_rows) > 10


def test_gpu_edges_example_and_state_errors(example_reads):
    R = _mdbg()
    with R.Mdbg(7, 10, 0.0008, 2) as m:
        m.ingest_reads(example_reads, 0)
        with pytest.raises(R.MdbgError) as ei:                   # ingested but not finalized
            m.graph_edges(0.01)
        assert ei.value.code == -6
        m.finalize()
        e = m.graph_edges(0.01)
        assert len(e["n1"]) == 206                               # tests/golden/example_cfg1.json
        m.ingest_reads(example_reads[:10], len(example_reads))   # the table changed: edges need a new finalize
        with pytest.raises(R.MdbgError):
            m.graph_edges(0.01)
    with R.Mdbg(7, 10, 0.0008, 2) as m:
        assert len(m.graph_edges(0.01)["n1"]) == 0               # empty context: empty list, no error
