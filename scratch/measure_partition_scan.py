"""cost of the ownership scan in the replicated-sketch mode: one context owning 1/W of the keys inserts a full config-3 sketch"""
import sys, json, time
sys.path.insert(0, '/root/repo')
import rust_mdbg_amd as R
n_reads = 466666
rows = []
for W in (1, 8):
    with R.Mdbg(35, 12, 0.002, 2, device=0) as m:
        if W > 1:
            m.set_partition(W, 0)
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
        m.sketch_device(db, do, n_reads, nb, 0)
        for rep in range(3):
            m.reset(35)                     # keeps the sketch, clears the table, re-inserts
            t = time.perf_counter(); m.reset(35); dt = time.perf_counter() - t
            st = m.stats()
        rows.append(dict(world=W, ms_insert_kernels=st["ms_insert"], ms_reset_incl_count=dt * 1e3, windows_owned=st["n_windows"]))
print(json.dumps(rows))
