"""stage times of one bench-shaped step (configs[2] shard) read from mdbg_get_stats; for the MDBG_INSERT_EXP / layout experiments
(no graph check: the experiment switches make the table wrong on purpose)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rust_mdbg_amd as R
k, l, d, A = 35, 12, 0.002, 2
genome_len = 140_000_000; n_reads = int(140e6 * 50 / 15000.0)
m = R.Mdbg(k, l, d, A, device=0)
d_bases, d_off, n_bases = m.synth_reads_device(seed=1, genome_len=genome_len, n_reads=n_reads, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000, first_read=0)
words = torch.zeros((n_bases + 31) // 32 + 2, dtype=torch.int64, device="cuda")
exc = (torch.zeros(64, dtype=torch.int64, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda"))
torch.cuda.synchronize()
assert m.pack_device(d_bases, n_bases, words.data_ptr(), exc[0].data_ptr(), exc[1].data_ptr(), 64) == 0
best = None
for it in range(6):
    m.reset(0)
    s0 = m.stats()
    m.ingest_packed_device(words.data_ptr(), d_off, n_reads, n_bases, 0)
    n = m.finalize_device().n
    s1 = m.stats()
    t = {f: s1[f] - s0[f] for f in ("ms_sketch", "ms_insert", "ms_finalize")}
    if it >= 2 and (best is None or t["ms_insert"] < best["ms_insert"]): best = t
print(os.environ.get("MDBG_INSERT_EXP", "0"), "nodes", n, {a: round(b, 4) for a, b in best.items()})
