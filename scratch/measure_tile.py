"""tile kernel time (HIP events, mean of N sketch-only steps on the bench's packed input); with MDBG_STOP_PHASE=n the tiles stop after phase n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rust_mdbg_amd as R
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m = R.Mdbg(35, 12, 0.002, 2, device=0)
n_reads = 466666
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert m.pack_device(db, nb, words.data_ptr()) == 0
tot = 0.0
for it in range(N + 5):
    m.reset(0); m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
    if it >= 5: tot += m.stats()["ms_sketch_tile"]
print("MDBG_STOP_PHASE=%s tile kernel %.4f ms (mean of %d)" % (os.environ.get("MDBG_STOP_PHASE", "-"), tot / N, N))
