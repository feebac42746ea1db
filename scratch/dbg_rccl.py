import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import rust_mdbg_amd as R
from rust_mdbg_amd import dist as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
nreads = 466666
m = R.Mdbg(35, 12, 0.002, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nreads)
eng = D.GpuEngine(m, torch, dev)
comm = D.TorchDistComm(dist, torch, dev)
for trial in range(2):
    eng.reset()
    eng.sketch_device(db, do, nreads, nb, 0)
    recs, counts = eng.route_pack(1)
    torch.cuda.synchronize()
    ref = recs.clone(); torch.cuda.synchronize()
    print("records", recs.shape, recs.is_contiguous(), hex(recs.data_ptr()), "view equals clone:", bool(torch.equal(recs, ref)))
    recv = torch.empty_like(ref)
    dist.all_to_all_single(recv, recs, output_split_sizes=counts, input_split_sizes=counts)
    torch.cuda.synchronize()
    neq = (recv != ref).any(1).sum().item()
    print("trial", trial, "rows differing after all_to_all from CAI view:", neq)
    recv2 = torch.empty_like(ref)
    dist.all_to_all_single(recv2, ref, output_split_sizes=counts, input_split_sizes=counts)
    torch.cuda.synchronize()
    print("   rows differing after all_to_all from torch-owned clone:", (recv2 != ref).any(1).sum().item())
    r3, _ = comm.alltoallv(recs, counts); torch.cuda.synchronize()
    print('   chunked alltoallv rows differing:', (r3 != ref).any(1).sum().item())
    if neq:
        bad = (recv != ref).any(1).nonzero().flatten()
        print("   first bad rows:", bad[:5].tolist(), "last:", bad[-5:].tolist(), "n_rows", ref.shape[0])
dist.destroy_process_group()
