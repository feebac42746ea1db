"""which call leans on zero-filled allocations: MDBG_POISON=1 python scratch/dbg_poison.py   (the human workload of bench.py in small, a sync and a line after every call)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import rust_mdbg_amd as R
import bench
def say(*a): print(*a, flush=True)
m = R.Mdbg(35, 14, 0.003, 2)
say("created")
batches, keep, shard_reads, _ = bench.human_shards(m, torch, np, 40.0, 52.0, range(8))
m.sync(); say("shards ok", shard_reads, [b[3] for b in batches])
for step in range(3):
    m.reset(0); m.sync(); say("reset ok")
    for i, (b_in, b_off, b_reads, b_bases, b_first) in enumerate(batches):
        m.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first); m.sync(); say(" ingest", i, "ok", m.stats()["n_minimizers"])
    nd = m.finalize_device(); m.sync(); say("finalize ok", int(nd.n))
