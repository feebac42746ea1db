"""Multi-rank driver (rust_mdbg_amd/dist.py) on CPU: world_size-2 gloo processes and in-process thread ranks, with the
numpy stand-in engine.  The merged node table must equal the sequential reference semantics (the oracle)."""
import os
import socket
import sys
import threading

import numpy as np
import pytest
import torch

from conftest import ROOT, TESTS
from oracle import oracle as O
from rust_mdbg_amd import dist as D, synth

K, L, DENS, A = 6, 12, 0.004, 2


def workload():
    return synth.synth_reads(3, 120000, 160, mean_len=9000, sd_len=2000, min_len=2000, max_len=15000, err_ppm=1500)


def to_np(part):
    return {f: (v.cpu().numpy().view(np.uint64) if hasattr(v, "cpu") else v) for f, v in part.items()}


def check_against_oracle(parts, reads, k=K, l=L, d=DENS, a=A):
    tab = D.gather_node_table([to_np(p) for p in parts])
    g = O.Graph(k, l, d, a)
    b, o = O.concat_reads(reads)
    g.ingest(b, o)
    exp = g.finalize(with_edges=False)
    assert tab["n_nodes"] == exp["n_nodes"] and tab["n_nodes_before"] == exp["n_nodes_before"]
    assert np.array_equal(tab["row"], np.arange(exp["n_nodes"], dtype=np.uint64))
    assert np.array_equal(tab["keys"], exp["keys"])
    for f in ("index", "abundance", "seqlen", "reversed", "src_read", "src_start", "src_end"):
        assert np.array_equal(tab[f].astype(np.uint64), exp[f].astype(np.uint64)), f
    assert np.array_equal(tab["shift_full"], exp["shift_full"])


def run_rank(rank, world, comm, reads, out, batches_per_rank=2, mode="route"):
    from engine_numpy import NumpyEngine
    eng = NumpyEngine(K, L, DENS, A)
    drv = D.DistributedMdbg(eng, comm, torch) if mode == "route" else D.ReplicatedMdbg(eng, comm, torch)
    per = len(reads) // world
    lo, hi = rank * per, (len(reads) if rank == world - 1 else (rank + 1) * per)
    step = (hi - lo + batches_per_rank - 1) // batches_per_rank
    chunks = [O.concat_reads(reads[s:min(hi, s + step)]) + (s,) for s in range(lo, hi, step)]
    if mode == "replicate-pipelined":          # all chunks in flight, windows inserted once everything has arrived
        drv.ingest_host_chunks(chunks)
    else:
        for bb, oo, s in chunks:
            drv.ingest_host(bb, oo, s)
    if mode != "route" and world > 1:          # the window lists travelled with every foreign sketch (the protocol of the GPU engine)
        foreign = [b for b in eng.batches if not (lo <= b["first"] < hi)]
        assert foreign and all("list" in b for b in foreign)
    out[rank] = drv.finalize()


@pytest.mark.parametrize("mode", ["route", "replicate", "replicate-pipelined"])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_thread_ranks_numpy_engine(world, mode):
    reads = workload()
    tw = D.ThreadWorld(world)
    out = [None] * world
    errs = []

    def body(r):
        try:
            run_rank(r, world, D.ThreadComm(tw, r, torch), reads, out, mode=mode)
        except BaseException as e:           # noqa: BLE001
            errs.append(e)
            tw.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    check_against_oracle(out, reads)


def test_uneven_ranks_pipelined_rounds():
    """ranks with different numbers of chunks (one of them with no reads at all): every rank still takes part in every
    round of the pipelined replicated-sketch exchange"""
    from engine_numpy import NumpyEngine
    reads = workload()[:90]
    world = 3
    tw = D.ThreadWorld(world)
    out, errs = [None] * world, []
    shares = {0: [(0, 30), (30, 50), (50, 70)], 1: [(70, 90)], 2: []}

    def body(r):
        try:
            drv = D.ReplicatedMdbg(NumpyEngine(K, L, DENS, A), D.ThreadComm(tw, r, torch), torch)
            chunks = [O.concat_reads(reads[a:b]) + (a,) for a, b in shares[r]]
            drv.ingest_host_chunks(chunks + [None] * (3 - len(chunks)))
            out[r] = drv.finalize()
        except BaseException as e:           # noqa: BLE001
            errs.append(e)
            tw.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    check_against_oracle(out, reads)
    assert D.plan_chunks(np.array([0, 5, 9], dtype=np.uint64), 4, keep_empty=True)[-1][:2] == (2, 2)
    assert len(D.plan_chunks(np.array([0, 5, 9], dtype=np.uint64), 4)) == 2


def _gloo_worker(rank, world, port, tmpdir, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = [None] * world
    run_rank(rank, world, D.TorchDistComm(dist, torch, torch.device("cpu"), max_bytes=1 << 16), workload(), out, mode=mode)
    np.savez(os.path.join(tmpdir, "part%d.npz" % rank), **{f: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for f, v in out[rank].items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["route", "replicate", "replicate-pipelined"])
def test_gloo_world2(tmp_path, mode):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    parts = []
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "part%d.npz" % r))
        parts.append({f: (z[f].view(np.uint64) if z[f].dtype == np.int64 and z[f].ndim else z[f]) for f in z.files})
    for p in parts:
        for f in ("n_nodes", "n_nodes_before", "n_local"):
            p[f] = int(p[f])
    check_against_oracle(parts, workload())
