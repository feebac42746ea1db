"""Round-4 additions checked on the GPU: the alternative tile geometries of the sketch kernel (kept in the tree behind MDBG_TILE as the record of an
experiment that measured slower) still give the oracle's sketches and tables; the process-wide block cache hands memory back when asked."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import rust_mdbg_amd as R
from oracle import oracle as O
rng = np.random.default_rng(11)
def rd(n, hp):
    out = []
    while len(out) < n:
        c = "ACGT"[int(rng.integers(4))]
        out += [c] * (1 + (int(rng.integers(1, 40)) if rng.random() < hp else 0))
    return "".join(out[:n]).encode()
reads = [rd(int(rng.integers(200, 90000)), 0.05) for _ in range(60)] + [b"", b"ACGT" * 3, b"C" * 700, rd(300000, 0.02), b"ACGTNNNNACGT" * 50, rd(20000, 0.3)]
for (k, l, d, a) in ((7, 10, 0.01, 2), (21, 12, 0.003, 1), (5, 14, 0.05, 2), (4, 20, 0.2, 1)):
    bases, offs = O.concat_reads(reads)
    with R.Mdbg(k, l, d, a) as m:
        sk = m.sketch(bases, offs)
        m.ingest(bases, offs, 0)
        got = m.finalize()
        st = m.stats()
    es = O.sketch(bases, offs, l, d)
    for f in ("hashes", "pos", "off"):
        assert np.array_equal(sk[f], es[f]), (f, k, l, d, a)
    g = O.Graph(k, l, d, a)
    assert g.ingest(bases, offs) == 0
    exp = g.finalize(with_edges=False)
    assert st["tile_bases"] == %d, st["tile_bases"]
    assert int(got["n_nodes"]) == exp["n_nodes"] and int(got["n_nodes_before"]) == exp["n_nodes_before"], (k, l, d, a)
    for f in ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed"):
        assert np.array_equal(got[f], exp[f]), (f, k, l, d, a)
print("TILE_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("shape,stride", [("1x4", 8064), ("1x1", 8064), ("4", 32512)])
def test_tile_geometries_give_the_oracles_table(shape, stride):
    from rust_mdbg_amd import api
    if shape != "4" and not (api.load_library().mdbg_build_flags() & 1):
        pytest.skip("the wave-tile kernels are compiled only with -DMDBG_WAVE_TILES since round 5 (scratch/build_variant.sh wave_tiles -DMDBG_WAVE_TILES)")
    env = dict(os.environ, MDBG_TILE=shape)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, stride)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "TILE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_block_cache_keeps_and_releases():
    import numpy as np
    import rust_mdbg_amd as R
    R.api.release_cached_memory()
    rng = np.random.default_rng(3)
    reads = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=400000).tobytes() for _ in range(40)]
    from oracle import oracle as O
    bases, offs = O.concat_reads(reads)
    for _ in range(2):                         # the second context finds the first one's blocks
        with R.Mdbg(21, 12, 0.01, 1) as m:          # (random reads share nothing: min_abundance 1)
            m.ingest(bases, offs, 0)
            n = int(m.finalize()["n_nodes"])
    freed = R.api.release_cached_memory()
    assert n > 0 and freed >= (8 << 20) and R.api.release_cached_memory() == 0


@pytest.mark.gpu
def test_nothing_leans_on_zero_filled_allocations():
    """MDBG_POISON hands every device block out filled with 0xA5.  A batch of empty reads only has no tile, so no gather wrote its reads' offsets: they were what the
    allocation held — zeros from a fresh hipMalloc, garbage in a process that had freed memory before (found by scratch/fuzz_nodes_edges.py, seeds 70346 / 70670)."""
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import test_gpu_fuzz as F\n"
             "for s in (70346, 70670, 3, 11): F.test_fuzz_sketch_and_nodes(s)\n"
             "import numpy as np, rust_mdbg_amd as R\n"
             "from oracle import oracle as O\n"
             "with R.Mdbg(5, 10, 0.05, 1) as m:\n"
             "    b, o = O.concat_reads([b'', b'', b''])\n"
             "    sk = m.sketch(b, o); assert list(sk['off']) == [0, 0, 0, 0], sk['off']\n"
             "    m.ingest_reads([b'ACGTTGCA' * 40], 0); m.ingest_reads([b'', b''], 1); m.ingest_reads([b'ACGTTGCA' * 40], 3)\n"
             "    g = O.Graph(5, 10, 0.05, 1); bb, oo = O.concat_reads([b'ACGTTGCA' * 40, b'', b'', b'ACGTTGCA' * 40]); assert g.ingest(bb, oo) == 0\n"
             "    got, exp = m.finalize(), g.finalize(with_edges=False)\n"
             "    assert got['n_nodes'] == exp['n_nodes'] and np.array_equal(got['keys'], exp['keys']) and np.array_equal(got['abundance'], exp['abundance'])\n"
             "print('POISON_OK')\n") % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=dict(os.environ, MDBG_POISON="1"), timeout=900)
    assert r.returncode == 0 and "POISON_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
