import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import rust_mdbg_amd as R
from rust_mdbg_amd import dist as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for nreads in (2000, 20000, 100000, 466666):
    m = R.Mdbg(35, 12, 0.002, 2, device=0)
    db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nreads)
    m.ingest_device(db, do, nreads, nb, 0)
    n_local = m.finalize_device().n; st = m.stats()
    res = [("local", n_local, st["n_distinct"], st["n_windows"])]
    for name, comm in (("thread", D.ThreadComm(D.ThreadWorld(1), 0, torch)), ("rccl", D.TorchDistComm(dist, torch, dev))):
        eng = D.GpuEngine(m, torch, dev); eng.reset()
        drv = D.DistributedMdbg(eng, comm, torch)
        t = time.time(); drv.ingest_device(db, do, nreads, nb, 0); torch.cuda.synchronize(); t1 = time.time() - t
        f = drv.finalize(); st = m.stats()
        res.append((name, f["n_nodes"], st["n_distinct"], st["n_windows"], round(t1 * 1e3, 1), round(st["ms_insert"], 1)))
    print(nreads, nb, res, flush=True)
    m.close()
