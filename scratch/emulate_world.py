"""W ranks as threads on ONE GPU (ThreadComm: the exchange is device copies): total GPU work of a W-rank weak-scaling step.
Per-rank compute cost at world W ~= step time / W (the ranks share the GPU); the xGMI transfer is NOT part of it."""
import sys, time, json, threading
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rust_mdbg_amd as R
from rust_mdbg_amd import dist as D
W = int(sys.argv[1]); steps = 3; n_reads = 466666; chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
tw = D.ThreadWorld(W); res = [None] * W; errs = []
def body(r):
    try:
        with R.Mdbg(35, 12, 0.002, 2, device=0) as m:
            db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000 * W, n_reads=n_reads, first_read=r * n_reads)
            eng = D.GpuEngine(m, torch, dev); drv = D.ReplicatedMdbg(eng, D.ThreadComm(tw, r, torch), torch)
            offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
            plan = D.plan_chunks(offs, chunks, keep_empty=True); offs_t = eng._view(do, (n_reads + 1,))
            ts = []
            for s in range(steps + 1):
                tw.barrier.wait(); t = time.perf_counter()
                drv.reset(); drv.ingest_device_chunked(db, offs_t, plan, r * n_reads); n = drv.finalize_device_count()
                torch.cuda.synchronize(); tw.barrier.wait(); ts.append(time.perf_counter() - t)
            res[r] = (n, drv.last_local, ts[1:], m.stats())
    except BaseException as e:
        errs.append(e); tw.barrier.abort()
th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in th]; [t.join() for t in th]
if errs: raise errs[0]
ms = 1e3 * float(np.mean(res[0][2]))
print(json.dumps(dict(world=W, chunks=chunks, ms_per_step_all_ranks_on_one_gpu=ms, ms_per_rank_equiv=ms / W, nodes=res[0][0], local_sum=sum(x[1] for x in res),
                      rank0_stage_ms=dict(sketch=res[0][3]["ms_sketch"], insert=res[0][3]["ms_insert"], finalize=res[0][3]["ms_finalize"]))))
