"""The multi-GPU path's device code on ONE GPU: `world` ranks run as threads, each with its own libmdbg context, exchanging
through ThreadComm.  Exercises mdbg_route_pack / insert_records / routed_export / resolve_* / routed_keys bit-exactly
against the oracle (the real 8-GPU run differs only in the communicator: RCCL all_to_all_single)."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from test_distributed_cpu import check_against_oracle

pytestmark = pytest.mark.gpu


def run(world, reads, k, l, d, a, batches_per_rank=2, mode="route"):
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    tw = D.ThreadWorld(world)
    out, errs = [None] * world, []
    dev = torch.device("cuda", 0)
    def share(rank):                          # reads [lo, hi) of a rank in steps of `step`
        per = len(reads) // world
        lo, hi = rank * per, (len(reads) if rank == world - 1 else (rank + 1) * per)
        return lo, hi, max(1, (hi - lo + batches_per_rank - 1) // batches_per_rank)

    n_rounds = max(len(range(*share(r))) for r in range(world))

    def body(rank):
        try:
            with R.Mdbg(k, l, d, a, device=0) as m:
                eng = D.GpuEngine(m, torch, dev)
                comm = D.ThreadComm(tw, rank, torch)
                drv = D.DistributedMdbg(eng, comm, torch) if mode == "route" else D.ReplicatedMdbg(eng, comm, torch)
                lo, hi, step = share(rank)
                chunks = [O.concat_reads(reads[s:min(hi, s + step)]) + (s,) for s in range(lo, hi, step)]
                chunks += [O.concat_reads([]) + (hi,)] * (n_rounds - len(chunks))      # every rank takes part in every round
                if mode.startswith("replicate-pipelined"):
                    # all chunks in flight into reserved regions of the peers' sketch stores (mdbg_sketch_reserve / _commit);
                    # "-nosize": the store is NOT sized up front, so it has to grow mid-flight -> drain-and-retry path
                    drv.sized = mode.endswith("nosize")
                    drv.ingest_host_chunks(chunks)
                else:
                    for bb, oo, s in chunks:
                        drv.ingest_host(bb, oo, s)
                part = drv.finalize()
                out[rank] = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
        except BaseException as e:           # noqa: BLE001
            errs.append(e)
            tw.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    return out


@pytest.mark.parametrize("mode", ["route", "replicate", "replicate-pipelined", "replicate-pipelined-nosize"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_routed_path_matches_oracle(world, mode):
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(3, 120000, 160, mean_len=9000, sd_len=2000, min_len=2000, max_len=15000, err_ppm=1500)
    parts = run(world, reads, 6, 12, 0.004, 2, batches_per_rank=3 if "pipelined" in mode else 2, mode=mode)
    check_against_oracle(parts, reads, 6, 12, 0.004, 2)


@pytest.mark.parametrize("mode", ["route", "replicate"])
@pytest.mark.parametrize("k,l,d,a", [(21, 12, 0.003, 2), (35, 12, 0.002, 2), (4, 10, 0.01, 1), (5, 12, 0.01, 3), (4, 10, 0.02, 11)])
def test_routed_path_configs(k, l, d, a, mode):
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(k, 300000, 500, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000)
    parts = run(2, reads, k, l, d, a, mode=mode)
    check_against_oracle(parts, reads, k, l, d, a)
    assert sum(p["n_local"] for p in parts) == parts[0]["n_nodes"] > 50


def test_rccl_communicator_world1_chunked():
    """TorchDistComm over the real RCCL backend (one rank), with a tiny per-message limit so that the exchange is cut
    into many rounds; the routed result must equal the local path and the oracle."""
    import os
    import torch
    import torch.distributed as dist
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        reads = synth.synth_reads(11, 200000, 300, mean_len=12000, sd_len=1500, min_len=4000, max_len=20000, err_ppm=1000)
        k, l, d, a = 9, 12, 0.004, 2
        with R.Mdbg(k, l, d, a, device=0) as m:
            eng = D.GpuEngine(m, torch, dev)
            comm = D.TorchDistComm(dist, torch, dev, max_bytes=4096)
            drv = D.DistributedMdbg(eng, comm, torch)
            b, o = O.concat_reads(reads)
            drv.ingest_host(b, o, 0)
            part = drv.finalize()
            part = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
        check_against_oracle([part], reads, k, l, d, a)
        # replicated-sketch mode over the same communicator, batch cut into chunks on the device (one rank: no peers, but
        # the counts all-gather, the chunk offsets and the store sizing run as they do on 8 GPUs)
        with R.Mdbg(k, l, d, a, device=0) as m:
            rep = D.ReplicatedMdbg(D.GpuEngine(m, torch, dev), comm, torch)
            tb, to = torch.from_numpy(b).to(dev), torch.from_numpy(o.view(np.int64)).to(dev)
            rep.ingest_device_chunked(tb.data_ptr(), to, D.plan_chunks(o, 4), 0)
            part = rep.finalize()
            part = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
        check_against_oracle([part], reads, k, l, d, a)
        assert comm.allgather_i64([7, 8]) == [[7, 8]]
        # raw communicator check: chunked alltoallv is the identity for one rank
        x = torch.arange(100003 * 3, device=dev, dtype=torch.int64).reshape(100003, 3)
        y, rc = comm.alltoallv(x, [100003])
        assert rc == [100003] and torch.equal(x, y)
    finally:
        dist.destroy_process_group()


def test_chunked_two_context_pipeline_matches_oracle():
    """sketch store and table in two contexts, batch cut into chunks (unaligned read starts -> lead-in bytes), producer
    thread sketching ahead of the exchange: same node table"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D, synth
    reads = synth.synth_reads(21, 250000, 320, mean_len=11000, sd_len=2500, min_len=100, max_len=20000, err_ppm=1500)
    k, l, d, a = 7, 12, 0.004, 2
    dev = torch.device("cuda", 0)
    b, o = O.concat_reads(reads)
    with R.Mdbg(k, l, d, a, device=0) as ms, R.Mdbg(k, l, d, a, device=0) as mt:
        eng = D.GpuEngine(ms, torch, dev, table=mt)
        drv = D.DistributedMdbg(eng, D.ThreadComm(D.ThreadWorld(1), 0, torch), torch)
        tb = torch.from_numpy(b).to(dev)
        to = torch.from_numpy(o.view(np.int64)).to(dev)
        plan = D.plan_chunks(o, 5)
        assert len(plan) == 5 and any(int(o[r0]) % 16 for r0, _, _, _ in plan)
        drv.ingest_device_chunked(tb.data_ptr(), to, plan, 1000)
        part = drv.finalize()
        part = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
    g = O.Graph(k, l, d, a)
    g.ingest(b, o, 1000)
    exp = g.finalize(with_edges=False)
    tab = D.gather_node_table([{f: (v.numpy().view(np.uint64) if hasattr(v, "numpy") else v) for f, v in part.items()}])
    assert tab["n_nodes"] == exp["n_nodes"] and tab["n_nodes_before"] == exp["n_nodes_before"]
    assert np.array_equal(tab["keys"], exp["keys"])
    for f in ("index", "abundance", "seqlen", "reversed", "src_read", "src_start", "src_end"):
        assert np.array_equal(tab[f].astype(np.uint64), exp[f].astype(np.uint64)), f


def test_uneven_ranks_pipelined_rounds_gpu():
    """ranks with different numbers of chunks, one with no reads at all, through the zero-copy import path"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D, synth
    reads = synth.synth_reads(5, 150000, 120, mean_len=9000, sd_len=2000, min_len=2000, max_len=15000, err_ppm=1500)
    k, l, d, a = 6, 12, 0.004, 2
    world = 3
    tw = D.ThreadWorld(world)
    out, errs = [None] * world, []
    shares = {0: [(0, 40), (40, 70), (70, 90)], 1: [(90, 120)], 2: []}
    dev = torch.device("cuda", 0)

    def body(r):
        try:
            with R.Mdbg(k, l, d, a, device=0) as m:
                drv = D.ReplicatedMdbg(D.GpuEngine(m, torch, dev), D.ThreadComm(tw, r, torch), torch)
                chunks = [O.concat_reads(reads[x:y]) + (x,) for x, y in shares[r]]
                drv.ingest_host_chunks(chunks + [None] * (3 - len(chunks)))
                part = drv.finalize()
                out[r] = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
        except BaseException as e:           # noqa: BLE001
            errs.append(e)
            tw.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    check_against_oracle(out, reads, k, l, d, a)


@pytest.mark.parametrize("mode", ["route", "replicate", "replicate-pipelined"])
def test_abundance_wrap_in_the_multi_rank_modes(mode):
    """k-min-mers seen > 65536 + A times (u16 abundance wraps in the reference): the partitioned tables must report the same
    sighting as the sequential semantics — replicated mode re-scans the resident global sketch, routed mode its record arena"""
    from test_gpu_parity import _repeat_reads
    reads = _repeat_reads(31, 420, 190, 200)
    k, l, d, a = 3, 8, 0.05, 2
    parts = run(2, reads, k, l, d, a, batches_per_rank=2, mode=mode)
    check_against_oracle(parts, reads, k, l, d, a)
    assert sum(p["n_local"] for p in parts) == parts[0]["n_nodes"] >= 3


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_multi_rank_modes(seed):
    """random parameters x random read sets x random world size / mode / batching (uneven ranks, empty rounds, read sets
    without a single window included) against the oracle"""
    import random
    from test_gpu_fuzz import fuzz_reads
    rnd = random.Random(9000 + seed)
    k, l, d, a = rnd.choice([(2, 8, 0.03, 1), (3, 8, 0.05, 2), (5, 10, 0.01, 2), (7, 12, 0.008, 3), (21, 12, 0.004, 2), (4, 6, 0.05, 8)])
    reads = fuzz_reads(rnd, n_reads=rnd.randint(12, 150), genome_len=rnd.choice([300, 5000, 40000]), mean_len=rnd.choice([40, 400, 4000]),
                       err=rnd.choice([0.0, 0.02]), p_lower=0.0, p_n=rnd.choice([0.0, 0.2]), p_hp=rnd.choice([0.0, 0.02]))
    reads = [r.replace(b"n", b"N") for r in reads]
    world = rnd.choice([1, 2, 3, 5])
    mode = rnd.choice(["route", "replicate", "replicate-pipelined", "replicate-pipelined-nosize"])
    parts = run(world, reads, k, l, d, a, batches_per_rank=rnd.choice([1, 2, 3]), mode=mode)
    check_against_oracle(parts, reads, k, l, d, a)


def test_replicated_multik_on_resident_global_sketch():
    """mdbg_reset(k) on every rank of the replicated-sketch mode: the global sketch is resident everywhere, so a new k needs no
    exchange at all (the owner counts shipped with the sketches were for the old k and must not be trusted again)"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D, synth
    reads = synth.synth_reads(9, 120000, 150, mean_len=9000, sd_len=2000, min_len=2000, max_len=15000, err_ppm=1500)
    l, d, a = 12, 0.004, 2
    world = 2
    tw = D.ThreadWorld(world)
    out, errs = {}, []
    dev = torch.device("cuda", 0)

    def body(r):
        try:
            with R.Mdbg(6, l, d, a, device=0) as m:
                eng = D.GpuEngine(m, torch, dev)
                drv = D.ReplicatedMdbg(eng, D.ThreadComm(tw, r, torch), torch)
                per = len(reads) // world
                lo, hi = r * per, (len(reads) if r == world - 1 else (r + 1) * per)
                drv.ingest_host_chunks([O.concat_reads(reads[lo:(lo + hi) // 2]) + (lo,), O.concat_reads(reads[(lo + hi) // 2:hi]) + ((lo + hi) // 2,)])
                for k in (6, 9, 4):
                    if k != 6:
                        m.reset(k)
                    part = drv.finalize()
                    out[(k, r)] = {f: (v.cpu() if hasattr(v, "cpu") else v) for f, v in part.items()}
        except BaseException as e:           # noqa: BLE001
            errs.append(e)
            tw.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in (6, 9, 4):
        check_against_oracle([out[(k, r)] for r in range(world)], reads, k, l, d, a)
