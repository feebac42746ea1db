"""What ONE rank of an eight-rank step computes on its own batch, measured on one GPU (no peers, no exchange): a human shard (19.5 Gbases, k=35 l=14 d=0.003) sketched in two
chunks, the owner lists for eight ranks, the insertion of the rank's own windows (one in eight, found by insert_windows_kernel itself), the partitioned finalize.  The figures
feed the time budget of DESIGN.md 3.4.  Second part: the RECEIVER side of the same step — the seven peers' shards are sketched one after the other by a second context of this
process, their owner lists taken, and rank 0 imports every peer's sketch (whole hashes, device-to-device copy standing in for the exchange) with the list of the windows it
owns and inserts them: what a rank of eight spends on the 7/8 of its table that come from its peers.  Missing from a real rank's step: the exchange itself.
usage: python scratch/measure_rank_w8.py [world] [genome_mb]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rust_mdbg_amd as R

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
genome_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
k, l, d, A = 35, 14, 0.003, 2
shard_reads = int(genome_mb * 1e6 * 52.0 / 15000.0) // 8
m = R.Mdbg(k, l, d, A, device=0)
m.set_partition(W, 0)
db, do, nb = m.synth_reads_device(seed=1, genome_len=int(genome_mb * 1e6), n_reads=shard_reads, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000, first_read=0)
words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert m.pack_device(db, nb, words.data_ptr()) == 0
m.sync()


def step():
    t = {}
    m.reset(0)
    m.sync(); t0 = time.perf_counter()
    m.ingest_packed_device(words.data_ptr(), do, shard_reads, nb, 0, sketch_only=True)
    m.sync(); t["sketch"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cnt, _ = m.owner_lists(W)
    m.sync(); t["owner_lists"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.insert_resident()
    m.sync(); t["insert_own"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    a, b, nw = m.finalize_begin()
    m.sync(); t["finalize_begin"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    nd, row, ng = m.finalize_end()
    m.sync(); t["finalize_end"] = time.perf_counter() - t0
    return t, cnt, int(nd.n), nw


best = None
for rep in range(4):
    t, cnt, n_nodes, nw = step()
    if rep and (best is None or sum(t.values()) < sum(best.values())):
        best = t
st = m.stats()
print("W=%d shard %.2f Gbases, %d minimizers, windows per owner max/mean %.3f, own windows %d, nodes of this rank %d, bitmap words %d" % (
    W, nb / 1e9, st["n_minimizers"], max(cnt) / (sum(cnt) / W), cnt[0], n_nodes, nw))
print("ms per stage (host time around each call, best of 3):", {k_: round(v * 1e3, 3) for k_, v in best.items()}, "sum %.3f" % (sum(best.values()) * 1e3))
print("library timers of the last pass:", {f: round(st[f], 3) for f in ("ms_sketch", "ms_sketch_tile", "ms_insert", "ms_finalize")})


# ---- the segments of this batch: what the layer's sender packs for seven peers, and what a receiver scatters (the same volume comes in as goes out) --------------------
import ctypes as C
m.reset(0)
m.ingest_packed_device(words.data_ptr(), do, shard_reads, nb, 0, sketch_only=True)
cnt, d_lists = m.owner_lists(W)
out = (C.c_double * 4)()
cc = (C.c_uint64 * W)(*cnt)
m.L.mdbg_dbg_segments_ms.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_double)]
rc = m.L.mdbg_dbg_segments_ms(m.h, W, 0, cc, d_lists, out)
print("segments: rc %d, %d list entries (own bucket of %d ships nothing), %d hashes packed (%.2f per shipped window, %.1f %% of the sketch): sender %.3f ms (counts, prefix, "
      "one host round trip, pack), receiver %.3f ms (counts, prefix, scatter)" % (rc, int(out[2]), cnt[0], int(out[3]), out[3] / max(1, out[2] - cnt[0]),
                                                                                 100.0 * out[3] / st["n_minimizers"], out[0], out[1]), flush=True)

# ---- the receiver side: rank 0 inserts its windows of the seven peers' sketches --------------------------------------------------------------------------------
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
peer = R.Mdbg(k, l, d, A, device=0)
peers = []
for r in range(1, W):          # the peers' sketches and lists are made once and kept (device copies of what the exchange would deliver)
    db2, do2, nb2 = peer.synth_reads_device(seed=1, genome_len=int(genome_mb * 1e6), n_reads=shard_reads, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000, first_read=r * shard_reads)
    w2 = torch.zeros((nb2 + 31) // 32 + 2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert peer.pack_device(db2, nb2, w2.data_ptr()) == 0
    peer.reset(0)
    peer.ingest_packed_device(w2.data_ptr(), do2, shard_reads, nb2, r * shard_reads, sketch_only=True)
    cnt2, d_lists = peer.owner_lists(W)
    peer.sync()
    bi = peer.last_batch()
    nm, nr = int(bi.n_minimizers), int(bi.n_reads)
    hs = torch.empty(nm, dtype=torch.int64, device="cuda"); ro = torch.empty(nr + 1, dtype=torch.int64, device="cuda"); ls = torch.empty(cnt2[0], dtype=torch.int64, device="cuda")
    assert hip.hipMemcpy(hs.data_ptr(), bi.d_hashes, nm * 8, 3) == 0
    assert hip.hipMemcpy(ro.data_ptr(), bi.d_read_offsets, (nr + 1) * 8, 3) == 0
    assert hip.hipMemcpy(ls.data_ptr(), d_lists, cnt2[0] * 8, 3) == 0              # bucket 0 comes first in the lists
    peers.append((r, nm, nr, hs, ro, ls, cnt2[0]))
    del w2, db2, do2
peer.close()


def receive():
    m.reset(0)
    m.ingest_packed_device(words.data_ptr(), do, shard_reads, nb, 0, sketch_only=True)
    m.owner_lists(W)
    n_listed = 0
    for r, nm, nr, hs, ro, ls, c0 in peers:
        dh, dp, region = m.sketch_reserve(nm)
        assert hip.hipMemcpy(dh, hs.data_ptr(), nm * 8, 3) == 0                       # (the exchange's stand-in)
        m.sketch_commit_listed(region, nm, ro.data_ptr(), nr, r * shard_reads, ls.data_ptr(), c0)
        n_listed += c0
    hip.hipDeviceSynchronize()                                                     # (hipMemcpy device-to-device returns before the copy has run: without this the copies ran under the insertion)
    m.sync(); t0 = time.perf_counter()
    m.insert_resident()                                                            # the rank's own windows and the listed ones in one round, as in a step of the multi-GPU layer
    m.sync(); t_ins = time.perf_counter() - t0
    t0 = time.perf_counter()
    a, b, nw = m.finalize_begin()
    m.sync(); t_fb = time.perf_counter() - t0
    t0 = time.perf_counter()
    nd, row, ng = m.finalize_end()
    m.sync(); t_fe = time.perf_counter() - t0
    return n_listed, t_ins, t_fb, t_fe, nw, int(nd.n)


for rep in range(3):               # (the first pass grows the table; the later ones find it large enough, as every step after a job's first does)
    n_listed, t_ins, t_fb, t_fe, nw, n_nodes = receive()
    st = m.stats()
    print("receiver pass %d [MDBG_LISTED_SPAN_MIN=%s]: own windows + %d listed windows of %d peers inserted in %.3f ms (library timer ms_insert %.3f); finalize over the whole "
          "index space (%d bitmap words): begin %.3f ms, end %.3f ms (no all-reduce, no position fetch), nodes of this rank %d, distinct keys %d" % (
              rep, os.environ.get("MDBG_LISTED_SPAN_MIN", "-"), n_listed, W - 1, t_ins * 1e3, st["ms_insert"], nw, t_fb * 1e3, t_fe * 1e3, n_nodes,
              st["n_distinct"]), flush=True)
