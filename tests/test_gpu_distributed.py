"""The multi-GPU path's device code on ONE GPU: `world` ranks run as threads, each with its own libmdbg context, exchanging
through ThreadComm.  Exercises mdbg_route_pack / insert_records / routed_export / resolve_* / routed_keys bit-exactly
against the oracle (the real 8-GPU run differs only in the communicator: RCCL all_to_all_single)."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from test_distributed_cpu import check_against_oracle

pytestmark = pytest.mark.gpu


def run(world, reads, k, l, d, a, batches_per_rank=2):
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import dist as D
    tw = D.ThreadWorld(world)
    out, errs = [None] * world, []
    dev = torch.device("cuda", 0)

    def body(rank):
        try:
            with R.Mdbg(k, l, d, a, device=0) as m:
                eng = D.GpuEngine(m, torch, dev)
                drv = D.DistributedMdbg(eng, D.ThreadComm(tw, rank, torch), torch)
                per = len(reads) // world
                lo, hi = rank * per, (len(reads) if rank == world - 1 else (rank + 1) * per)
                step = (hi - lo + batches_per_rank - 1) // batches_per_rank
                for s in range(lo, hi, step):
                    bb, oo = O.concat_reads(reads[s:min(hi, s + step)])
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


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_routed_path_matches_oracle(world):
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(3, 120000, 160, mean_len=9000, sd_len=2000, min_len=2000, max_len=15000, err_ppm=1500)
    parts = run(world, reads, 6, 12, 0.004, 2)
    check_against_oracle(parts, reads, 6, 12, 0.004, 2)


@pytest.mark.parametrize("k,l,d,a", [(21, 12, 0.003, 2), (35, 12, 0.002, 2), (4, 10, 0.01, 1), (5, 12, 0.01, 3)])
def test_routed_path_configs(k, l, d, a):
    from rust_mdbg_amd import synth
    reads = synth.synth_reads(k, 300000, 500, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000)
    parts = run(2, reads, k, l, d, a)
    check_against_oracle(parts, reads, k, l, d, a)
    assert sum(p["n_local"] for p in parts) == parts[0]["n_nodes"] > 50
