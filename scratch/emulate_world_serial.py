"""W ranks as threads on ONE GPU with every GPU phase SERIALISED (a global lock around each engine call, device sync inside it):
the time a rank spends in each phase is then its isolated GPU cost, not inflated by seven other ranks' kernels sharing the chip.
The exchange itself (device copies here, xGMI on a node) is timed separately and NOT charged to the rank: bytes are reported.
usage: emulate_world_serial.py <W> [chunks] [config: 2|3]"""
import sys, time, json, threading, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rust_mdbg_amd as R
from rust_mdbg_amd import dist as D
W = int(sys.argv[1]); chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1; cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = 3
if cfg == 2: k, l, d, n_reads, glen = 35, 12, 0.002, 466666, 140_000_000
else: k, l, d, n_reads, glen = 35, 14, 0.003, 1_300_000, 375_000_000
packed = True
dev = torch.device("cuda", 0)
gpu = threading.Lock()
PH = ["sketch_device", "owner_lists", "store_reserve", "reserve_import", "commit_import", "insert_owned", "finalize_begin", "finalize_end", "reset"]


class TimedEngine(D.GpuEngine):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw); self.tm_ = collections.defaultdict(float)

def _wrap(name):
    base = getattr(D.GpuEngine, name)
    def f(self, *a, **kw):
        with gpu:
            torch.cuda.synchronize(); t = time.perf_counter()
            r = base(self, *a, **kw)
            torch.cuda.synchronize(); self.tm_[name] += time.perf_counter() - t
        return r
    return f
for name in PH: setattr(TimedEngine, name, _wrap(name))


class TimedComm(D.ThreadComm):
    def __init__(self, *a):
        super().__init__(*a); self.t_x = 0.0; self.t_ar = 0.0; self.bytes = 0
    def exchange(self, sends, recvs):
        allv = self._exchange_keep(dict((peer, ts) for peer, ts in sends))
        with gpu:
            torch.cuda.synchronize(); t = time.perf_counter()
            for peer, ts in recvs:
                for dst, src in zip(ts, allv[peer][self.rank]):
                    if dst.numel(): dst.copy_(src); self.bytes += dst.numel() * dst.element_size()
            torch.cuda.synchronize(); self.t_x += time.perf_counter() - t
        self.tw.barrier.wait()
        return D._Pending([], None, None)
    def allreduce_sum_(self, x):
        allv = self._exchange(x.clone())
        with gpu:
            torch.cuda.synchronize(); t = time.perf_counter()
            x.copy_(sum(allv[1:], allv[0]))
            torch.cuda.synchronize(); self.t_ar += time.perf_counter() - t
        self.tw.barrier.wait()
        return x

tw = D.ThreadWorld(W); res = [None] * W; errs = []
def body(r):
    try:
        with R.Mdbg(k, l, d, 2, device=0) as m:
            db, do, nb = m.synth_reads_device(seed=1, genome_len=glen * W, n_reads=n_reads, first_read=r * n_reads)
            eng = TimedEngine(m, torch, dev)
            if packed:
                words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
                ep = torch.zeros(16, dtype=torch.int64, device="cuda"); ev = torch.zeros(16, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()          # the fills run on torch's stream, the packer on the context's
                assert m.pack_device(db, nb, words.data_ptr(), ep.data_ptr(), ev.data_ptr(), 16) == 0
                eng.packed = True; src = words.data_ptr()
            else: src = db
            comm = TimedComm(tw, r, torch); drv = D.ReplicatedMdbg(eng, comm, torch)
            offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
            plan = D.plan_chunks(offs, chunks, keep_empty=True); offs_t = eng._view(do, (n_reads + 1,))
            for s in range(steps + 1):
                if s == 1: eng.tm_.clear(); comm.t_x = comm.t_ar = 0.0; comm.bytes = 0
                tw.barrier.wait()
                drv.reset(); drv.ingest_device_chunked(src, offs_t, plan, r * n_reads); n = drv.finalize_device_count()
                torch.cuda.synchronize(); tw.barrier.wait()
            res[r] = (n, nb, dict((p, 1e3 * v / steps) for p, v in eng.tm_.items()), 1e3 * comm.t_x / steps, 1e3 * comm.t_ar / steps, comm.bytes / steps)
    except BaseException as e:
        errs.append(e); tw.barrier.abort()
th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in th]; [t.join() for t in th]
if errs: raise errs[0]
ph = dict((p, float(np.mean([x[2].get(p, 0.0) for x in res]))) for p in PH)
comp = sum(ph.values())
print(json.dumps(dict(world=W, chunks=chunks, config=cfg, bases_per_rank=res[0][1], nodes=res[0][0], per_rank_compute_ms=round(comp, 3),
                      phases_ms=dict((p, round(v, 3)) for p, v in ph.items() if v > 0.0005),
                      allreduce_emulated_ms=round(float(np.mean([x[4] for x in res])), 3),
                      exchange_copy_ms_not_charged=round(float(np.mean([x[3] for x in res])), 3),
                      exchange_bytes_received_per_rank=int(np.mean([x[5] for x in res])))))
