import sys, json
sys.path.insert(0, '.')
import rust_mdbg_amd as R
n_reads = 133333
out = []
for (l, s, d) in ((12, 4, 0.05), (14, 6, 0.05), (12, 0, 0.005)):
    m = R.Mdbg(10, l, d, 2, syncmer_s=s)
    db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
    for rep in range(2):
        m.reset(0); m.sketch_device(db, do, n_reads, nb, 0)
    st = m.stats()
    out.append(dict(l=l, s=s, density=d, gbases=nb / 1e9, ms_kernel=st["ms_sketch_tile"], ms_sketch=st["ms_sketch"], gbases_per_s=nb / (st["ms_sketch_tile"] * 1e-3) / 1e9, minimizers=st["n_minimizers"]))
    m.close()
print(json.dumps(out))
