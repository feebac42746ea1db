"""sketch stage at dense settings (tile kernel / whole stage, packed input): l = 12, d = 0.03 and 0.1 (the reference CLI's default density), 2.0 Gbases"""
import sys
sys.path.insert(0, '.')
import torch
import rust_mdbg_amd as R
n_reads = 133333
for d in (0.002, 0.03, 0.1):
    m = R.Mdbg(10, 12, d, 2)
    db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
    words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert m.pack_device(db, nb, words.data_ptr()) == 0
    for rep in range(3):
        m.reset(0); m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
    st = m.stats()
    print("d=%.3f tile %.3f ms (%.2f Tbases/s), sketch stage %.3f ms, minimizers %d" % (d, st["ms_sketch_tile"], nb / st["ms_sketch_tile"] / 1e9, st["ms_sketch"], st["n_minimizers"]))
    m.close()
