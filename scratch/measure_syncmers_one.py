"""one syncmer configuration, one sketch launch after a warm-up (for rocprofv3 counter passes): usage: python scratch/measure_syncmers_one.py [l] [s] [density]"""
import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_mdbg_amd as R
l, s, d = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 4, float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
n_reads = 133333
m = R.Mdbg(10, l, d, 2, syncmer_s=s)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
for rep in range(2):
    m.reset(0); m.sketch_device(db, do, n_reads, nb, 0)
st = m.stats()
print(json.dumps(dict(l=l, s=s, density=d, gbases=nb / 1e9, ms_kernel=st["ms_sketch_tile"], gbases_per_s=nb / (st["ms_sketch_tile"] * 1e-3) / 1e9)))
