"""tile-kernel time with ASCII vs packed input on the bench workload (one GPU)"""
import sys, json
sys.path.insert(0, '.')
import torch
import rust_mdbg_amd as R
k, l, d, a = 35, 12, 0.002, 2
if len(sys.argv) > 1: l = int(sys.argv[1])
if len(sys.argv) > 2: d = float(sys.argv[2])
n_reads = 466666
m = R.Mdbg(k, l, d, a)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
ep = torch.zeros(16, dtype=torch.int64, device="cuda"); ev = torch.zeros(16, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
assert m.pack_device(db, nb, words.data_ptr(), ep.data_ptr(), ev.data_ptr(), 16) == 0
out = {}
for name in ("ascii", "packed", "ascii", "packed"):
    m.reset(0)
    if name == "ascii": m.sketch_device(db, do, n_reads, nb, 0)
    else: m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
    st = m.stats()
    out[name] = dict(ms_tile=st["ms_sketch_tile"], ms_sketch=st["ms_sketch"], minimizers=st["n_minimizers"], slow=st["n_slow_tiles"])
mpb = out["ascii"]["minimizers"] / nb
for name, bin_ in (("ascii", 1.0), ("packed", 0.25)):
    o = out[name]; o["GBps"] = nb * (bin_ + 12 * mpb) / (o["ms_tile"] * 1e-3) / 1e9; o["frac_of_8TBps"] = o["GBps"] / 8000; o["Tbases_per_s"] = nb / (o["ms_tile"] * 1e-3) / 1e12
print(json.dumps(dict(l=l, d=d, n_bases=nb, **out)))
