"""Round-6 additions checked on the GPU."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_parity import _mdbg, assert_nodes_equal, oracle_graph


@pytest.mark.gpu
@pytest.mark.parametrize("no_claims", [False, True])
def test_a_node_that_stops_being_solid_between_two_finalize_calls(no_claims):
    """The abundance is a u16 that wraps (src/main.rs:663, release build) and the filter looks at the wrapped value (:927), so a k-min-mer can be solid at one finalize
    and not at the next: 65,535 sightings (abundance 65535: solid for minabund 2), one more (65536 = u16 0: gone), two more (u16 2: back).  The claim-map finalize
    keeps "solid" as bit 1 of the claim byte and until round 6 only ever SET it, so the second finalize listed no solid slot but left the bits in the bitmap
    (round-5 advice).  Child process: MDBG_NO_CLAIMS (the byte-map path, which zeroes its maps per finalize) is read per finalize, the check is the same."""
    child = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import random
import numpy as np
from test_gpu_parity import _mdbg, assert_nodes_equal, oracle_graph
rnd = random.Random(3)
read = bytes(rnd.choice(b"ACGT") for _ in range(420))
other = [bytes(rnd.choice(b"ACGT") for _ in range(900)) for _ in range(6)]      # keys seen twice from the start: their claim bytes carry both bits all along
k, l, d, A = 3, 8, 0.05, 2
reads = other + other + [read] * 65535
R = _mdbg()
with R.Mdbg(k, l, d, A) as m:
    m.ingest_reads(reads, 0)
    a = m.finalize()
    exp_a = oracle_graph(reads, k, l, d, A)
    assert_nodes_equal(a, exp_a)
    n_a = exp_a["n_nodes"]
    assert int(np.max(exp_a["abundance"])) == 65535
    m.ingest_reads([read], len(reads)); reads = reads + [read]
    b = m.finalize()
    exp_b = oracle_graph(reads, k, l, d, A)
    assert exp_b["n_nodes"] < n_a and exp_b["n_nodes"] >= 6, (exp_b["n_nodes"], n_a)          # the wrapped keys left the table, the others stay
    assert_nodes_equal(b, exp_b)
    assert_nodes_equal(m.finalize(), exp_b)
    m.ingest_reads([read, read], len(reads)); reads = reads + [read, read]
    c = m.finalize()
    exp_c = oracle_graph(reads, k, l, d, A)
    assert exp_c["n_nodes"] == n_a
    assert_nodes_equal(c, exp_c)
    e = m.graph_edges(0.01)
print("WRAP_BETWEEN_OK", n_a, exp_b["n_nodes"])
""" % (ROOT, ROOT)
    env = dict(os.environ)
    if no_claims:
        env["MDBG_NO_CLAIMS"] = "1"
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "WRAP_BETWEEN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_nodes_digest_of_the_device_table_equals_the_oracles():
    """mdbg_nodes_digest (include/mdbg_hip.h) over the device table of mdbg_finalize_device = the oracle's digest of its own node table (plain numpy over the oracle's
    rows, and orc_count_digest_threaded over the reads), for minabund 1 and 2; an empty table digests to (0, 0)"""
    from oracle import oracle as O
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(11, 200000, 500, mean_len=9000, sd_len=1500, min_len=3000, max_len=15000, err_ppm=1500)
    b, o = O.concat_reads(reads)
    R = _mdbg()
    for (k, l, d, A) in ((9, 12, 0.004, 2), (5, 10, 0.01, 1), (21, 12, 0.003, 2)):
        exp = oracle_graph(reads, k, l, d, A)
        want = O.nodes_digest(exp["keys"], exp["abundance"])
        assert O.count_digest_threaded(b, o, k, l, d, A, threads=4)[2] == want
        with R.Mdbg(k, l, d, A) as m:
            m.ingest(b, o, 0)
            nd = m.finalize_device()
            assert int(nd.n) == exp["n_nodes"] > 100 and m.nodes_digest(nd) == want
            m.reset(0)
            assert m.nodes_digest(m.finalize_device()) == (0, 0)


@pytest.mark.gpu
def test_segment_and_stage_timer_hooks_answer():
    """the two measurement entry points of round 6 (include/mdbg_hip.h: mdbg_dbg_segments_ms; include/mdbg_dist.h: mdbg_dist_stage_ms) on a small batch: the counts they
    report are the owner lists' own, every shipped window adds between 1 and k hashes, and the timers are finite"""
    import ctypes as C
    R = _mdbg()
    W, k, l, d = 4, 21, 12, 0.01
    with R.Mdbg(k, l, d, 2) as m:
        m.set_partition(W, 1)
        db, do, nb = m.synth_reads_device(seed=4, genome_len=2_000_000, n_reads=3000)
        m.sketch_device(db, do, 3000, nb, 0)
        cnt, d_lists = m.owner_lists(W)
        assert sum(cnt) > 10000 and min(cnt) > 0
        out = (C.c_double * 4)()
        m.L.mdbg_dbg_segments_ms.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_double)]
        assert m.L.mdbg_dbg_segments_ms(m.h, W, 1, (C.c_uint64 * W)(*cnt), d_lists, out) == 0
        shipped = sum(cnt) - cnt[1]
        assert int(out[2]) == sum(cnt) and shipped <= int(out[3]) <= shipped * k and 0 < out[0] < 1000 and 0 < out[1] < 1000
        assert m.L.mdbg_dbg_segments_ms(m.h, 65, 0, (C.c_uint64 * W)(*cnt), d_lists, out) != 0      # more than 64 ranks: refused
