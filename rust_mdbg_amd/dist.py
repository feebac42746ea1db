"""Multi-GPU driver: one process per GPU, reads sharded by record, one all-to-all of k-min-mer records per batch.

The reference is a single process (threads + one DashMap, src/main.rs:595,834); this is the MI355X counterpart of that
shared map: every k-min-mer occurrence is routed to the rank that owns its key range (owner = mulhi64(keyhash, world)),
with `all_to_all_single` over RCCL/xGMI (torch.distributed backend "nccl"), and the two order-dependent fields of a
node — DbgEntry.index (order of first sighting) and the A-th sighting's seqlen/shift — are resolved by asking the rank
that generated the read in question (two more, small, all-to-alls at finalize).

The driver only moves tensors; all compute is in the engine (GpuEngine = libmdbg_hip.so through its C ABI).
Communicators: TorchDistComm (RCCL on GPUs, gloo on CPU) and ThreadComm (several ranks inside one process, for tests).
All tensors are int64 (u64 values bit-cast); ~0 appears as -1.
"""
import threading

WIN_BITS = 26


# ------------------------------------------------------------------------------------------- communicators
class TorchDistComm:
    """torch.distributed communicator (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU).

    Large exchanges are cut into rounds of at most `max_bytes` per (source, destination) message: RCCL 2.26 was
    observed to deliver only the first ~0.95 GB of a 1.9 GB all_to_all_single message (profiles/r01_notes.md), and
    bounded rounds also bound the staging memory of the collective."""

    def __init__(self, dist, torch, device, max_bytes=256 << 20):
        self.dist, self.torch, self.device, self.max_bytes = dist, torch, device, max_bytes
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def alltoallv(self, send, counts):
        """send: [n, ...] rows grouped by destination; counts[d] rows go to rank d -> (recv rows, recv counts)"""
        t, dist = self.torch, self.dist
        counts = [int(c) for c in counts]
        sc = t.tensor(counts, dtype=t.int64, device=self.device)
        rc = t.empty_like(sc)
        dist.all_to_all_single(rc, sc)
        rcl = [int(x) for x in rc.tolist()]
        send = send.contiguous()
        recv = t.empty((sum(rcl),) + tuple(send.shape[1:]), dtype=send.dtype, device=self.device)
        row_bytes = max(1, send.element_size() * (send.numel() // max(1, send.shape[0]) if send.shape[0] else 1))
        max_rows = max(1, self.max_bytes // row_bytes)
        biggest = t.tensor([max(counts + rcl + [0])], dtype=t.int64, device=self.device)
        dist.all_reduce(biggest, op=dist.ReduceOp.MAX)
        rounds = (int(biggest.item()) + max_rows - 1) // max_rows
        if rounds <= 1:
            dist.all_to_all_single(recv, send, output_split_sizes=rcl, input_split_sizes=counts)
            return recv, rcl
        so, ro = [0], [0]
        for c in counts:
            so.append(so[-1] + c)
        for c in rcl:
            ro.append(ro[-1] + c)
        for r in range(rounds):
            a = r * max_rows
            ins = [send[so[d] + min(a, counts[d]): so[d] + min(a + max_rows, counts[d])] for d in range(self.world)]
            outs = [recv[ro[s] + min(a, rcl[s]): ro[s] + min(a + max_rows, rcl[s])] for s in range(self.world)]
            i_sp, o_sp = [int(x.shape[0]) for x in ins], [int(x.shape[0]) for x in outs]
            sbuf = t.cat(ins, 0) if self.world > 1 else ins[0]
            rbuf = t.empty((sum(o_sp),) + tuple(send.shape[1:]), dtype=send.dtype, device=self.device) if self.world > 1 else outs[0]
            dist.all_to_all_single(rbuf, sbuf, output_split_sizes=o_sp, input_split_sizes=i_sp)
            if self.world > 1:
                o = 0
                for s in range(self.world):
                    outs[s].copy_(rbuf[o:o + o_sp[s]])
                    o += o_sp[s]
        return recv, rcl

    def allgather_obj(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class ThreadWorld:
    """shared state of `world` in-process ranks (one thread each)"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.box = [None] * world


class ThreadComm:
    def __init__(self, tw, rank, torch):
        self.tw, self.rank, self.world, self.torch = tw, rank, tw.world, torch

    def _exchange(self, obj):
        self.tw.box[self.rank] = obj
        self.tw.barrier.wait()
        allv = list(self.tw.box)
        self.tw.barrier.wait()
        return allv

    def alltoallv(self, send, counts):
        t = self.torch
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + int(c))
        allv = self._exchange((send, offs))
        parts, rc = [], []
        for src in range(self.world):
            s, o = allv[src]
            parts.append(s[o[self.rank]:o[self.rank + 1]])
            rc.append(o[self.rank + 1] - o[self.rank])
        return t.cat(parts, 0).clone(), rc

    def allgather_obj(self, obj):
        return self._exchange(obj)


# ------------------------------------------------------------------------------------------- GPU engine
class _DevArray:
    """zero-copy view of library-owned device memory for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


class GpuEngine:
    """libmdbg_hip.so behind the stage interface the driver needs"""

    def __init__(self, mdbg, torch, device):
        self.m, self.t, self.device, self.ranges = mdbg, torch, device, []

    @property
    def k(self):
        return self.m.k

    def _view(self, ptr, shape):
        n = 1
        for s in shape:
            n *= s
        if n == 0 or not ptr:
            return self.t.empty(shape, dtype=self.t.int64, device=self.device)
        return self.t.as_tensor(_DevArray(ptr, shape), device=self.device)

    def reset(self):
        self.m.reset(0)
        self.ranges = []

    def sketch_device(self, d_bases, d_offsets, n_reads, n_bases, first_ordinal):
        self.m.sketch_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal)
        self.ranges.append((int(first_ordinal), int(n_reads)))

    def sketch_host(self, bases, offsets, first_ordinal):
        # host buffers: stage through the library, sketch only
        import numpy as np
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        tb = self.t.from_numpy(bases).to(self.device) if len(bases) else self.t.empty(16, dtype=self.t.uint8, device=self.device)
        to = self.t.from_numpy(offsets.view(np.int64)).to(self.device)
        self.t.cuda.synchronize()
        self.sketch_device(tb.data_ptr(), to.data_ptr(), len(offsets) - 1, int(offsets[-1]), first_ordinal)

    def route_pack(self, world):
        ptr, counts = self.m.route_pack(world)
        return self._view(ptr, (sum(counts), self.k + 1)), counts

    def insert_records(self, recs):
        recs = recs.contiguous()
        self.t.cuda.synchronize()
        self.m.insert_records(recs.data_ptr() if recs.numel() else 0, recs.shape[0])

    def export(self):
        n, a, b, c, d = self.m.routed_export()
        first, ath, slot = self._view(a, (n,)), self._view(b, (n,)), self._view(d, (n,))
        cnt = self.t.as_tensor(_DevArray(c, ((n + 1) // 2,)), device=self.device).view(self.t.int32)[:n].to(self.t.int64) if n else self.t.empty(0, dtype=self.t.int64, device=self.device)
        return first, ath, cnt, slot

    def resolve_first(self, ords, solid):
        t = self.t
        n = ords.shape[0]
        rf = t.empty(n, dtype=t.int64, device=self.device)
        rs = t.empty(n, dtype=t.int64, device=self.device)
        ords = ords.contiguous()
        solid = solid.to(t.uint8).contiguous()
        t.cuda.synchronize()
        tf, ts = self.m.resolve_first(ords.data_ptr() if n else 0, solid.data_ptr() if n else 0, n, rf.data_ptr() if n else 0, rs.data_ptr() if n else 0)
        return rf, rs, tf, ts

    def resolve_meta(self, ords):
        t = self.t
        n = ords.shape[0]
        meta = t.empty((n, 6), dtype=t.int64, device=self.device)
        ords = ords.contiguous()
        t.cuda.synchronize()
        if n:
            self.m.resolve_meta(ords.data_ptr(), n, meta.data_ptr())
        return meta

    def keys(self, slots):
        t = self.t
        n = slots.shape[0]
        out = t.empty((n, self.k), dtype=t.int64, device=self.device)
        slots = slots.contiguous()
        t.cuda.synchronize()
        if n:
            self.m.routed_keys(slots.data_ptr(), n, out.data_ptr())
        return out


# ------------------------------------------------------------------------------------------- driver
class DistributedMdbg:
    def __init__(self, engine, comm, torch):
        self.e, self.c, self.t = engine, comm, torch

    # process_read_aux over this rank's shard of a batch (src/main.rs:730-785), distributed
    def ingest_device(self, d_bases, d_offsets, n_reads, n_bases, first_ordinal):
        self.e.sketch_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal)
        self.exchange()

    def ingest_host(self, bases, offsets, first_ordinal):
        self.e.sketch_host(bases, offsets, first_ordinal)
        self.exchange()

    def exchange(self):
        recs, counts = self.e.route_pack(self.c.world)
        recv, _ = self.c.alltoallv(recs, counts)
        self.e.insert_records(recv)

    def _generator_of(self, ords, all_ranges):
        """rank that sketched the read an ordinal belongs to"""
        t = self.t
        flat = sorted((fo, n, r) for r, rl in enumerate(all_ranges) for (fo, n) in rl)
        starts = t.tensor([f[0] for f in flat], dtype=t.int64, device=ords.device)
        owner = t.tensor([f[2] for f in flat], dtype=t.int64, device=ords.device)
        idx = t.searchsorted(starts, ords >> WIN_BITS, right=True) - 1
        return owner[idx.clamp(min=0)]

    def _query(self, payload, dest):
        """send rows of payload to dest ranks, return (recv rows, recv counts, order) — replies go back with _reply"""
        t = self.t
        order = t.argsort(dest, stable=True)
        counts = t.bincount(dest, minlength=self.c.world).tolist()
        recv, rc = self.c.alltoallv(payload[order], counts)
        return recv, rc, order, counts

    def _reply(self, answer, rc, order, counts):
        back, _ = self.c.alltoallv(answer, rc)
        out = self.t.empty_like(back)
        out[order] = back
        return out

    def finalize(self):
        """-> this rank's partition of the node table (dict of tensors) plus global counters; rows carry their global
        `row` (= position in index order), so concatenating all partitions and sorting by row gives the reference's table"""
        t, e, c = self.t, self.e, self.c
        first, ath, count, slot = e.export()
        all_ranges = c.allgather_obj(e.ranges)
        spans = [(min(fo for fo, _ in rl), max(fo + n for fo, n in rl)) if rl else None for rl in all_ranges]
        live = sorted((s[0], r) for r, s in enumerate(spans) if s)
        for (a, ra), (b, rb) in zip(live, live[1:]):
            assert spans[ra][1] <= b, "each rank must hold one contiguous range of read ordinals"
        solid = ath != -1
        # 1. first sightings -> index and row
        q = t.stack([first, solid.to(t.int64)], 1)
        recv, rc, order, counts = self._query(q, self._generator_of(first, all_ranges))
        rf, rs, tf, ts = e.resolve_first(recv[:, 0].contiguous(), recv[:, 1].contiguous())
        totals = c.allgather_obj((tf, ts))
        base_f = base_s = 0
        for _, r in live:
            if r == c.rank:
                break
            base_f += totals[r][0]
            base_s += totals[r][1]
        ans = self._reply(t.stack([rf + base_f, rs + base_s], 1), rc, order, counts)
        index_all, row_all = ans[:, 0], ans[:, 1]
        # 2. metadata of the A-th sighting, solid nodes only
        sel = solid.nonzero().flatten()
        a_sel = ath[sel]
        recv2, rc2, order2, counts2 = self._query(a_sel, self._generator_of(a_sel, all_ranges))
        meta = self._reply(e.resolve_meta(recv2.contiguous()), rc2, order2, counts2)
        keys = e.keys(slot[sel])
        n_nodes = sum(x[1] for x in totals)
        n_before = sum(x[0] for x in totals)
        return dict(keys=keys, index=index_all[sel], row=row_all[sel], abundance=count[sel] & 0xFFFF, seqlen=meta[:, 0] & 0xFFFFFFFF,
                    reversed=(meta[:, 0] >> 32) & 1, shift_full=meta[:, 1:3], src_read=meta[:, 3], src_start=meta[:, 4], src_end=meta[:, 5],
                    n_nodes=n_nodes, n_nodes_before=n_before, n_local=int(sel.shape[0]))

    def finalize_device_count(self):
        return self.finalize()["n_nodes"]


def gather_node_table(parts):
    """test helper: list of per-rank finalize() dicts (numpy-converted) -> one table sorted by row"""
    import numpy as np
    cat = lambda f: np.concatenate([np.asarray(p[f]) for p in parts], 0)
    row = cat("row")
    o = np.argsort(row, kind="stable")
    out = {f: cat(f)[o] for f in ("keys", "index", "abundance", "seqlen", "reversed", "shift_full", "src_read", "src_start", "src_end")}
    out["row"] = row[o]
    out["n_nodes"] = parts[0]["n_nodes"]
    out["n_nodes_before"] = parts[0]["n_nodes_before"]
    return out
