"""The GPU path against the committed fixtures of the independent restatement (tests/golden/independent_cases.json): sketches, node
tables field by field, edges (GPU edge builder) and the alphabet error — ASCII and 2-bit packed input."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "independent_cases.json")))["cases"]


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_equals_independent_fixture(ci, packed):
    import rust_mdbg_amd as R
    from rust_mdbg_amd import emit as E
    from oracle import oracle as O
    c = CASES[ci]
    reads = [r.encode() for r in c["reads"]]
    bases, offs = O.concat_reads(reads)
    with R.Mdbg(c["k"], c["l"], c["density"], c["minabund"], reads_already_hpc=c["already_hpc"]) as m:
        try:
            if packed:
                m.ingest_packed(E.pack_reads(bases, offs), 0)
            else:
                m.ingest(bases, offs, 0)
        except R.MdbgError as ex:
            assert ex.code == -2 and "error_read" in c and ("read %d " % c["error_read"]) in str(ex)
            return
        assert "error_read" not in c
        sk = m.store_sketch()
        r = m.finalize()
        ge = m.graph_edges(c["presimp"])
    o = sk["off"]
    for i, (pos, hs) in enumerate(c["sketch"]):
        assert sk["pos"][int(o[i]):int(o[i + 1])].tolist() == pos and sk["hashes"][int(o[i]):int(o[i + 1])].tolist() == hs, ("sketch of read", i)
    assert r["n_nodes"] == c["n_nodes"] and r["n_nodes_before"] == c["n_nodes_before"]
    for row, n in enumerate(c["nodes"]):
        assert r["keys"][row].tolist() == n["key"] and int(r["index"][row]) == n["index"] and int(r["abundance"][row]) == n["abundance"]
        assert int(r["seqlen"][row]) == n["seqlen"] and r["shift"][row].tolist() == n["shift"] and r["shift_full"][row].tolist() == n["shift_full"]
        assert int(r["reversed"][row]) == n["reversed"] and int(r["src_read"][row]) == n["src_read"]
        assert int(r["src_start"][row]) == n["src_start"] and int(r["src_end"][row]) == n["src_end"]
    got = sorted([int(a), chr(b), int(cc), chr(d), int(e)] for a, b, cc, d, e in zip(ge["n1"], ge["o1"], ge["n2"], ge["o2"], ge["overlap"]))
    assert got == sorted(c["edges"]) and ge["presimp_removed"] == c["presimp_removed"]
