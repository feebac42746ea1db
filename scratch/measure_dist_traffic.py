"""W ranks of the C layer (include/mdbg_dist.h) as THREADS on one GPU with a thread communicator written here (device copies for the exchange):
bytes a rank receives per step, segments vs whole sketches, and the GPU time of the step (all ranks share the GPU: per-rank cost ~ time / W;
the link time is NOT part of it).  usage: measure_dist_traffic.py <W> <config: 2 | 3> [chunks]"""
import sys, os, json, threading, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_mdbg_amd as R
from rust_mdbg_amd import api, dist_c

W = int(sys.argv[1]); cfg = int(sys.argv[2]); chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if cfg == 2: k, l, d, genome, n_reads = 35, 12, 0.002, 140_000_000, 466666
else: k, l, d, genome, n_reads = 35, 14, 0.003, 375_000_000, 1300000
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
D2D, D2H, H2D = 3, 2, 1

class Xfer(C.Structure):
    _fields_ = [("peer", C.c_uint32), ("d_ptr", C.c_void_p), ("bytes", C.c_uint64)]

bar = threading.Barrier(W)
ag = [None] * W; posted = [None] * W; red = [None] * W
AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64))
EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_uint32, C.POINTER(Xfer), C.c_uint32)
AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)

def make_comm(rank):
    def allgather(_, send, n, recv):
        ag[rank] = [send[i] for i in range(n)]
        bar.wait()
        for r in range(W):
            for i in range(n): recv[r * n + i] = ag[r][i]
        bar.wait()
        return 0
    def exchange(_, sends, ns, recvs, nr):
        posted[rank] = [(sends[i].peer, sends[i].d_ptr, sends[i].bytes) for i in range(ns)]
        bar.wait()
        nxt = {}
        for i in range(nr):
            p = recvs[i].peer
            mine = [x for x in posted[p] if x[0] == rank]
            j = nxt.get(p, 0); nxt[p] = j + 1
            assert mine[j][2] == recvs[i].bytes, (rank, p, j, mine[j][2], recvs[i].bytes)
            if hip.hipMemcpy(recvs[i].d_ptr, mine[j][1], recvs[i].bytes, D2D) != 0: return -4
        if hip.hipDeviceSynchronize() != 0: return -4
        bar.wait()
        return 0
    def allreduce(_, d_buf, n):
        a = np.empty(n, dtype=np.uint64)
        if n and hip.hipMemcpy(a.ctypes.data, d_buf, n * 8, D2H) != 0: return -4
        red[rank] = a
        bar.wait()
        s = red[0].copy()
        for r in range(1, W): s += red[r]
        bar.wait()
        if n and hip.hipMemcpy(d_buf, s.ctypes.data, n * 8, H2D) != 0: return -4
        return 0
    fns = (AG(allgather), EX(exchange), AR(allreduce))
    cm = dist_c.Comm(); cm.self = None; cm.rank = rank; cm.world = W
    cm.allgather_u64 = C.cast(fns[0], C.c_void_p); cm.exchange = C.cast(fns[1], C.c_void_p); cm.allreduce_sum_u64 = C.cast(fns[2], C.c_void_p)
    return cm, fns

L = api.load_library()
L.mdbg_dist_create.restype = C.c_void_p; L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(dist_c.Comm), C.POINTER(C.c_int)]
L.mdbg_dist_ingest_batch_packed_device.argtypes = [C.c_void_p, C.POINTER(api.PackedBatch), C.c_uint64, C.c_uint64]
L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
L.mdbg_dist_reset.argtypes = [C.c_void_p, C.c_uint32]; L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]; L.mdbg_dist_set_exchange.argtypes = [C.c_void_p, C.c_uint32]
L.mdbg_dist_destroy.argtypes = [C.c_void_p]; L.mdbg_dist_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
L.mdbg_dist_ctx.restype = C.c_void_p; L.mdbg_dist_ctx.argtypes = [C.c_void_p]
res = {}; errs = []

def body(rank):
    try:
        import torch
        cm, keep = make_comm(rank)
        P = api.Params(k=k, l=l, density=d, min_abundance=2, reads_already_hpc=0, device=0, flags=0, table_capacity_hint=0)
        err = C.c_int()
        h = L.mdbg_dist_create(C.byref(P), C.byref(cm), C.byref(err)); assert h, err.value
        with R.Mdbg(k, l, d, 2, device=0) as gen:
            db, do, nb = gen.synth_reads_device(seed=1, genome_len=genome * W, n_reads=n_reads, first_read=rank * n_reads)
            words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
            exc = (torch.zeros(64, dtype=torch.int64, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda"))
            torch.cuda.synchronize()
            assert gen.pack_device(db, nb, words.data_ptr(), exc[0].data_ptr(), exc[1].data_ptr(), 64) == 0
            pb = api.PackedBatch(words.data_ptr(), do, n_reads, 0, 0, 0)
            out = {}
            for whole in ((0,) if os.environ.get("MODES") == "segments" else (1,) if os.environ.get("MODES") == "whole" else (0, 1)):
                assert L.mdbg_dist_set_exchange(h, whole) == 0 and L.mdbg_dist_set_pipeline(h, chunks) == 0
                ts = []
                for step in range(3):
                    bar.wait(); t0 = time.perf_counter()
                    assert L.mdbg_dist_reset(h, 0) == 0
                    e = L.mdbg_dist_ingest_batch_packed_device(h, C.byref(pb), nb, rank * n_reads); assert e == 0, e
                    nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
                    e = L.mdbg_dist_finalize(h, C.byref(nd), C.byref(row), C.byref(ng)); assert e == 0, e
                    torch.cuda.synchronize(); bar.wait(); ts.append(time.perf_counter() - t0)
                a, b, q = C.c_uint64(), C.c_uint64(), C.c_uint64()
                L.mdbg_dist_traffic(h, C.byref(a), C.byref(b), C.byref(q))
                out["whole" if whole else "segments"] = dict(bytes_in_per_step=int(a.value), bytes_out_per_step=int(b.value), position_queries=int(q.value), nodes_global=int(ng.value), nodes_local=int(nd.n),
                                                             ms_per_step_all_ranks_on_one_gpu=1e3 * float(np.mean(ts[1:])))
            res[rank] = out
        L.mdbg_dist_destroy(h)
    except BaseException as e:
        errs.append(e); bar.abort(); raise

th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in th]; [t.join() for t in th]
if errs: raise errs[0]
summ = {"world": W, "config": cfg, "chunks": chunks, "k": k, "l": l, "density": d, "reads_per_rank": n_reads}
for mode in [m_ for m_ in ("segments", "whole") if m_ in res[0]]:
    bi = [res[r][mode]["bytes_in_per_step"] for r in range(W)]
    summ[mode] = dict(bytes_in_per_rank_per_step_max=max(bi), bytes_in_per_rank_per_step_mean=float(np.mean(bi)), nodes_global=res[0][mode]["nodes_global"],
                      nodes_local=[res[r][mode]["nodes_local"] for r in range(W)], ms_per_step_all_ranks_on_one_gpu=res[0][mode]["ms_per_step_all_ranks_on_one_gpu"],
                      ms_per_rank_equiv=res[0][mode]["ms_per_step_all_ranks_on_one_gpu"] / W)
print(json.dumps(summ))
