"""Zero-copy sketch import (include/mdbg_hip.h: mdbg_store_reserve / mdbg_sketch_reserve / mdbg_sketch_commit / mdbg_last_batch):
a context that receives another context's sketch into reserved regions of its store must produce the node table the
oracle produces for the same reads; misuse is reported, not silently accepted."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import assert_nodes_equal, rand_reads

pytestmark = pytest.mark.gpu


def test_import_into_reserved_regions_matches_oracle():
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    k, l, d, a = 5, 12, 0.006, 2
    base = rand_reads(5, 40, 3000, 9000)
    reads = base + [r[100:] for r in base[:25]] + rand_reads(6, 10, 0, 40)       # repeats, so that some nodes are solid
    dev = torch.device("cuda", 0)
    cuts = [0, 20, 21, 50, len(reads)]
    with R.Mdbg(k, l, d, a, device=0) as src, R.Mdbg(k, l, d, a, device=0) as dst:
        es, ed = D.GpuEngine(src, torch, dev), D.GpuEngine(dst, torch, dev)
        # dst sketches batches 1 and 2 itself and receives 0 and 3 from src; regions are reserved before their data exists
        # and own sketching goes on in between, so batches are NOT adjacent in the store (unused boundary slots)
        pend = []

        def own(i):
            bb, oo = O.concat_reads(reads[cuts[i]:cuts[i + 1]])
            ed.sketch_host(bb, oo, cuts[i])

        def reserve(i):
            bb, oo = O.concat_reads(reads[cuts[i]:cuts[i + 1]])
            es.sketch_host(bb, oo, cuts[i])
            h, p, off, first, n = es.last_sketch()
            assert first == cuts[i] and n == cuts[i + 1] - cuts[i]
            (hv, pv, token), = ed.reserve_import([h.shape[0]])
            pend.append((hv, pv, token, h.clone(), p.clone(), off.clone(), first))

        dst.store_reserve(1 << 20, 1 << 12)
        own(1); reserve(0); own(2); reserve(3)
        with pytest.raises(R.MdbgError) as ei:             # uncommitted regions: insertion is refused
            dst.insert_resident()
        assert ei.value.code == -6
        for hv, pv, token, h, p, off, first in pend:
            hv.copy_(h); pv.copy_(p)
            torch.cuda.synchronize()
            ed.commit_import(token, off, first)
        ed.insert_owned()
        got = dst.finalize()
        st = dst.stats()
    g = O.Graph(k, l, d, a)
    bb, oo = O.concat_reads(reads)
    assert g.ingest(bb, oo) == 0
    exp = g.finalize(with_edges=False)
    assert exp["n_nodes"] > 20
    assert_nodes_equal(got, exp)
    assert st["n_windows"] > 0


def test_import_misuse_is_reported():
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    dev = torch.device("cuda", 0)
    reads = rand_reads(9, 12, 4000, 6000)
    bb, oo = O.concat_reads(reads)
    with R.Mdbg(4, 12, 0.01, 1, device=0) as src, R.Mdbg(4, 12, 0.01, 1, device=0) as dst:
        es, ed = D.GpuEngine(src, torch, dev), D.GpuEngine(dst, torch, dev)
        es.sketch_host(bb, oo, 0)
        h, p, off, first, n = es.last_sketch()
        (hv, pv, token), = ed.reserve_import([h.shape[0]])
        # the store must not move while a region is pending: a reservation far beyond its capacity is refused, not served
        with pytest.raises(R.MdbgError) as ei:
            dst.sketch_reserve(1 << 28)
        assert ei.value.code == -6
        hv.copy_(h); pv.copy_(p)
        bad = off.clone(); bad[-1] += 1                     # offsets that do not end at the region size
        torch.cuda.synchronize()
        ed.commit_import(token, bad, 0)
        with pytest.raises(R.MdbgError) as ei:
            ed.insert_owned()
        assert ei.value.code == -1
    with R.Mdbg(4, 12, 0.01, 1, device=0) as dst:
        with pytest.raises(R.MdbgError):                    # nothing sketched or imported yet
            dst.last_batch()
        with pytest.raises(R.MdbgError):                    # commit of something that was never reserved
            dst.sketch_commit(0, 10, 0, 1, 0)


def test_copying_import_sketch_view_ingest_sketch():
    """mdbg_sketch_view + mdbg_ingest_sketch (the copying variant of the import): context B built only from A's sketch"""
    import rust_mdbg_amd as R
    k, l, d, a = 4, 12, 0.008, 1
    reads = rand_reads(21, 30, 2000, 7000) + [b"", b"ACGT"]
    bb, oo = O.concat_reads(reads)
    with R.Mdbg(k, l, d, a, device=0) as src, R.Mdbg(k, l, d, a, device=0) as dst:
        src.ingest(bb, oo, 0)
        v = src.sketch_view()
        assert int(v.n_reads) == len(reads)
        dst.ingest_sketch(v.d_hashes, v.d_positions, v.d_read_offsets, int(v.n_reads), 0)
        dst.insert_resident()
        got = dst.finalize()
    g = O.Graph(k, l, d, a)
    assert g.ingest(bb, oo) == 0
    assert_nodes_equal(got, g.finalize(with_edges=False))


def test_owner_counts_are_used_and_verified():
    """mdbg_owner_counts: the sender's per-owner window counts size the receivers' tables without a re-count; a count that
    does not match the imported sketch is reported by the insertion, not silently accepted"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    dev = torch.device("cuda", 0)
    k, l, d, a = 5, 12, 0.006, 2
    base = rand_reads(61, 40, 3000, 9000)
    reads = base + [r[50:] for r in base[:30]]
    bb, oo = O.concat_reads(reads)
    for wrong in (False, True):
        with R.Mdbg(k, l, d, a, device=0) as src, R.Mdbg(k, l, d, a, device=0) as d0, R.Mdbg(k, l, d, a, device=0) as d1:
            es = D.GpuEngine(src, torch, dev)
            es.sketch_host(bb, oo, 0)
            counts = src.owner_counts(2)
            assert len(counts) == 2 and min(counts) > 0
            h, p, off, first, n = es.last_sketch()
            parts = []
            for rank, dst in enumerate((d0, d1)):
                dst.set_partition(2, rank)
                ed = D.GpuEngine(dst, torch, dev)
                (hv, pv, token), = ed.reserve_import([h.shape[0]])
                hv.copy_(h); pv.copy_(p)
                torch.cuda.synchronize()
                ed.commit_import(token, off.clone(), first, counts[rank] + (7 if wrong and rank == 1 else 0))
                if wrong and rank == 1:
                    with pytest.raises(R.MdbgError) as ei:
                        ed.insert_owned()
                    assert ei.value.code == -1
                    continue
                ed.insert_owned()
                assert dst.stats()["n_windows"] == counts[rank]
                bf, bs = ed.finalize_begin()
                parts.append((ed, bf, bs))
            if wrong:
                continue
            tot_f, tot_s = parts[0][1] + parts[1][1], parts[0][2] + parts[1][2]      # the driver's all-reduce
            for ed, bf, bs in parts:
                bf.copy_(tot_f); bs.copy_(tot_s)
            torch.cuda.synchronize()
            outs = [ed.finalize_end() for ed, _, _ in parts]
            tab = D.gather_node_table([{f: (v.cpu().numpy().view(np.uint64) if hasattr(v, "cpu") else v) for f, v in o.items()} for o in outs])
        g = O.Graph(k, l, d, a)
        assert g.ingest(bb, oo) == 0
        exp = g.finalize(with_edges=False)
        assert tab["n_nodes"] == exp["n_nodes"] > 20 and np.array_equal(tab["keys"], exp["keys"])
        assert np.array_equal(tab["abundance"].astype(np.uint64), exp["abundance"].astype(np.uint64))
