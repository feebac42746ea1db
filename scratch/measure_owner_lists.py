"""time of mdbg_owner_lists (the sender's share of a round besides the sketch) on the bench shard, and the windows per owner"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rust_mdbg_amd as R
m = R.Mdbg(35, 12, 0.002, 2, device=0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=466666)
m.sketch_device(db, do, 466666, nb, 0)
for W in (2, 8):
    ts = []
    for _ in range(4):
        m.sync(); t0 = time.perf_counter(); cnt, _p = m.owner_lists(W); m.sync(); ts.append(time.perf_counter() - t0)
    print("W=%d owner_lists %.3f ms; windows per owner max/mean %.3f %s" % (W, 1e3 * min(ts), max(cnt) / (sum(cnt) / W), cnt))
