"""Multi-GPU driver: one process per GPU, reads sharded by record, one all-to-all of k-min-mer records per batch.

The reference is a single process (threads + one DashMap, src/main.rs:595,834); this is the MI355X counterpart of that
shared map: every k-min-mer occurrence is routed to the rank that owns its key range (owner = mulhi64(keyhash, world)),
with `all_to_all_single` over RCCL/xGMI (torch.distributed backend "nccl"), and the two order-dependent fields of a
node — DbgEntry.index (order of first sighting) and the A-th sighting's seqlen/shift — are resolved by asking the rank
that generated the read in question (two more, small, all-to-alls at finalize).

The driver only moves tensors; all compute is in the engine (GpuEngine = libmdbg_hip.so through its C ABI).
Communicators: TorchDistComm (RCCL on GPUs, gloo on CPU) and ThreadComm (several ranks inside one process, for tests).
All tensors are int64 (u64 values bit-cast); ~0 appears as -1.  A routed record is k+2 values: key[k], ordinal, key hash.
"""
import ctypes as C
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

    def alltoallv(self, send, counts, alloc=None):
        """send: [n, ...] rows grouped by destination; counts[d] rows go to rank d -> (recv rows, recv counts).
        alloc(n_rows), if given, provides the receive tensor (e.g. a view of the engine's record arena: no copy later)"""
        t, dist = self.torch, self.dist
        counts = [int(c) for c in counts]
        sc = t.tensor(counts, dtype=t.int64, device=self.device)
        rc = t.empty_like(sc)
        dist.all_to_all_single(rc, sc)
        rcl = [int(x) for x in rc.tolist()]
        send = send.contiguous()
        recv = alloc(sum(rcl)) if alloc is not None else t.empty((sum(rcl),) + tuple(send.shape[1:]), dtype=send.dtype, device=self.device)
        row_bytes = send.element_size()              # from the trailing dimensions only: identical on every rank, also for an empty send
        for dim in send.shape[1:]:
            row_bytes *= int(dim)
        row_bytes = max(1, row_bytes)
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

    def allreduce_sum_(self, x):
        self.dist.all_reduce(x)
        return x

    def allgather_i64(self, vals):
        """small fixed-size all-gather of integers -> [world][len(vals)]"""
        t = self.torch
        x = t.tensor([int(v) for v in vals], dtype=t.int64, device=self.device)
        out = t.empty(self.world * x.shape[0], dtype=t.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, x)
        return out.view(self.world, -1).tolist()

    def exchange(self, sends, recvs):
        """Point-to-point exchange in one group (RCCL send/recv pairs use the direct xGMI link of every pair of GPUs):
        sends / recvs = [(peer, [tensors])], the tensors of a (source, destination) pair are matched in order; empty
        tensors are skipped on both sides.  Returns a handle; wait() blocks the host until every receive has landed."""
        dist, ops = self.dist, []

        def pieces(x):          # both sides cut a tensor the same way: at most max_bytes per message (see the class comment)
            rows = max(1, self.max_bytes // max(1, x.element_size() * (x.numel() // max(1, x.shape[0]))))
            return [x[a:a + rows] for a in range(0, x.shape[0], rows)] if x.numel() else []
        for peer, ts in recvs:
            ops += [dist.P2POp(dist.irecv, y, peer) for x in ts for y in pieces(x)]
        for peer, ts in sends:
            ops += [dist.P2POp(dist.isend, y, peer) for x in ts for y in pieces(x)]
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return _Pending(reqs, self.torch if self.device is not None and getattr(self.device, "type", "cpu") == "cuda" else None, (sends, recvs))


class _Pending:
    def __init__(self, reqs, torch_cuda, keep):
        self.reqs, self.t, self.keep = reqs, torch_cuda, keep      # keep: the tensors stay referenced until wait()

    def wait(self):
        for r in self.reqs:
            r.wait()
        if self.t is not None and self.reqs:
            self.t.cuda.current_stream().synchronize()           # Work.wait() only orders the current stream; the engine has its own
        self.reqs, self.keep = [], None


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

    def alltoallv(self, send, counts, alloc=None):
        t = self.torch
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + int(c))
        allv = self._exchange_keep((send, offs))          # the senders' buffers (views of their engines' memory) stay untouched ...
        parts, rc = [], []
        for src in range(self.world):
            s, o = allv[src]
            parts.append(s[o[self.rank]:o[self.rank + 1]])
            rc.append(o[self.rank + 1] - o[self.rank])
        out = t.cat(parts, 0).clone()
        if out.is_cuda:
            t.cuda.current_stream().synchronize()
        self.tw.barrier.wait()                           # ... until every receiver has finished copying
        return out, rc

    def allgather_obj(self, obj):
        return self._exchange(obj)

    def allreduce_sum_(self, x):
        allv = self._exchange(x.clone())
        x.copy_(sum(allv[1:], allv[0]))
        return x


    def allgather_i64(self, vals):
        return [list(v) for v in self._exchange([int(v) for v in vals])]

    def exchange(self, sends, recvs):
        allv = self._exchange_keep(dict((peer, ts) for peer, ts in sends))
        for peer, ts in recvs:
            for dst, src in zip(ts, allv[peer][self.rank]):
                if dst.numel():
                    dst.copy_(src)
                    if dst.is_cuda:
                        self.torch.cuda.current_stream().synchronize()
        self.tw.barrier.wait()
        return _Pending([], None, None)

    def _exchange_keep(self, obj):
        """like _exchange but the senders' objects stay valid until the caller's closing barrier"""
        self.tw.box[self.rank] = obj
        self.tw.barrier.wait()
        return list(self.tw.box)


# ------------------------------------------------------------------------------------------- GPU engine
class _DevArray:
    """zero-copy view of library-owned device memory for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, shape, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class GpuEngine:
    """libmdbg_hip.so behind the stage interface the driver needs"""

    def __init__(self, mdbg, torch, device, table=None):
        """mdbg: context holding the sketch store (sketch / route / resolve); table: optional SECOND context for the
        owner side (arena / insert / export / keys).  In routed mode the two sides share no state, so giving them their own
        contexts (own HIP streams) lets DistributedMdbg overlap one chunk's sketch with the previous chunk's exchange."""
        self.m, self.t, self.device, self.ranges = mdbg, torch, device, []
        self.tm = table if table is not None else mdbg
        self.packed = False              # True: sketch_device gets 2-bit packed words (mdbg_packed_batch layout) instead of ASCII

    @property
    def k(self):
        return self.m.k

    def _view(self, ptr, shape, i32=False):
        n = 1
        for s in shape:
            n *= s
        if n == 0 or not ptr:
            return self.t.empty(shape, dtype=self.t.int32 if i32 else self.t.int64, device=self.device)
        return self.t.as_tensor(_DevArray(ptr, shape, "<i4" if i32 else "<i8"), device=self.device)

    def reset(self):
        self.m.reset(0)
        if self.tm is not self.m:
            self.tm.reset(0)
        self.ranges = []

    def sketch_device(self, d_bases, d_offsets, n_reads, n_bases, first_ordinal):
        if self.packed:
            self.m.ingest_packed_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal, sketch_only=True)
        else:
            self.m.sketch_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal)
        self.ranges.append((int(first_ordinal), int(n_reads)))

    def base_ptr(self, d_bases, a):
        """pointer to base a (a multiple of 64, see plan_chunks) of a device batch in the engine's input format"""
        return d_bases + (a // 4 if self.packed else a)

    def sketch_host(self, bases, offsets, first_ordinal):
        # host buffers: stage through the library, sketch only
        import numpy as np
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        tb = self.t.from_numpy(bases).to(self.device) if len(bases) else self.t.empty(16, dtype=self.t.uint8, device=self.device)
        to = self.t.from_numpy(offsets.view(np.int64)).to(self.device)
        self.t.cuda.synchronize()
        self.sketch_device(tb.data_ptr(), to.data_ptr(), len(offsets) - 1, int(offsets[-1]), first_ordinal)

    # replicated-sketch mode ------------------------------------------------------------------------------
    def set_partition(self, world, rank):
        self.m.set_partition(world, rank)

    def store_reserve(self, n_minimizers, n_reads):
        self.m.store_reserve(n_minimizers, n_reads)

    def last_sketch(self):
        """the batch sketched last -> (hashes view, positions view (int32), offsets relative to the batch, first ordinal, n_reads)"""
        b = self.m.last_batch()
        m, n = int(b.n_minimizers), int(b.n_reads)
        off = self._view(b.d_read_offsets, (n + 1,)) - int(b.store_offset)
        return self._view(b.d_hashes, (m,)), self._view(b.d_positions, (m,), i32=True), off, int(b.first_read_ordinal), n

    def reserve_import(self, sizes):
        """room for peers' sketches of sizes[i] minimizers at the end of the resident store (no copy after the receive)
        -> [(hashes view, positions view, token)]"""
        dh, dp, region = self.m.sketch_reserve(sum(sizes))
        out, o = [], 0
        for m in sizes:
            out.append((self._view(dh + 8 * o, (m,)), self._view(dp + 4 * o, (m,), i32=True), (region + o, m)))
            o += m
        return out

    def owner_counts(self, world):
        return self.m.owner_counts(world)

    def owner_lists(self, world):
        """-> (windows of the batch sketched last per owning rank, their lists bucketed by owner as one int32 view)"""
        counts, p = self.m.owner_lists(world)
        return counts, self._view(p, (2 * sum(counts),), i32=True)          # pairs (window, read) of int32

    def commit_import(self, token, rel_off, first_ordinal, owned=None, window_list=None):
        rel_off = rel_off.contiguous()
        self._keep = getattr(self, "_keep", [])
        self._keep.append(rel_off)             # the call is stream-ordered: keep the offsets alive until insert_owned()
        if window_list is not None:            # the sender listed this rank's windows (copied by the library)
            window_list = window_list.contiguous()
            self._keep.append(window_list)
            self.m.sketch_commit_listed(token[0], token[1], rel_off.data_ptr(), rel_off.shape[0] - 1, first_ordinal,
                                        window_list.data_ptr() if window_list.shape[0] else 0, int(window_list.shape[0]) // 2)
        else:
            self.m.sketch_commit(token[0], token[1], rel_off.data_ptr(), rel_off.shape[0] - 1, first_ordinal, owned)

    def insert_owned(self):
        self.m.insert_resident()
        self._keep = []

    def finalize_begin(self):
        a, b, n = self.m.finalize_begin()
        return self._view(a, (n,)), self._view(b, (n,))

    def finalize_end(self, counts_only=False):
        t = self.t
        nd, row, ng = self.m.finalize_end()
        n, k = int(nd.n), self.k
        if counts_only:                    # the node rows stay in the library's device buffers (mdbg_nodes of mdbg_finalize_end)
            return dict(n_nodes=int(ng), n_nodes_before=int(nd.n_distinct), n_local=n)
        addr = lambda p: C.cast(p, C.c_void_p).value or 0
        v16 = lambda p, cnt: self._view_as(addr(p), cnt, t.int16, 2) & 0xFFFF
        return dict(keys=self._view(addr(nd.keys), (n, k)), index=self._view_as(addr(nd.index), n, t.int32, 4) & 0xFFFFFFFF, row=self._view(row, (n,)),
                    abundance=v16(nd.abundance, n), seqlen=self._view_as(addr(nd.seqlen), n, t.int32, 4) & 0xFFFFFFFF,
                    reversed=self._view_as(addr(nd.reversed), n, t.uint8, 1), shift_full=self._view(addr(nd.shift_full), (n, 2)),
                    src_read=self._view(addr(nd.src_read), (n,)), src_start=self._view(addr(nd.src_start), (n,)), src_end=self._view(addr(nd.src_end), (n,)),
                    n_nodes=int(ng), n_nodes_before=int(nd.n_distinct), n_local=n)

    def route_pack(self, world):
        ptr, counts = self.m.route_pack(world)
        return self._view(ptr, (sum(counts), self.k + 2)), counts

    def route_empty(self, world):
        """what a rank with nothing to send contributes to a round"""
        return self._view(0, (0, self.k + 2)), [0] * world

    def alloc_records(self, n):
        """receive buffer for n routed records inside the library's arena (zero-copy insert)"""
        return self._view(self.tm.arena_reserve(n), (n, self.k + 2))

    def insert_records(self, recs):
        recs = recs.contiguous()
        self.t.cuda.synchronize()
        self.tm.insert_records(recs.data_ptr() if recs.numel() else 0, recs.shape[0])

    def _view_as(self, ptr, n, dtype, itemsize):
        """n elements of a narrower integer type at ptr, as an int64 tensor (copy)"""
        t = self.t
        if n == 0 or not ptr:
            return t.empty(0, dtype=t.int64, device=self.device)
        words = (n * itemsize + 7) // 8
        return self._view(ptr, (words,)).view(dtype)[:n].to(t.int64)

    def export(self, world, span_lo, span_rank):
        r = self.tm.routed_export(world, span_lo, span_rank)
        na, ns = int(r.n_all), int(r.n_solid)
        return dict(first=self._view(r.d_first, (na,)), solid=self._view_as(r.d_solid, na, self.t.uint8, 1),
                    counts_all=[int(r.counts_all[i]) for i in range(world)],
                    ath=self._view(r.d_ath, (ns,)), count=self._view_as(r.d_count, ns, self.t.int32, 4), slot=self._view(r.d_slot, (ns,)),
                    idx_all=self._view(r.d_idx_all, (ns,)), counts_solid=[int(r.counts_solid[i]) for i in range(world)])

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
            self.tm.routed_keys(slots.data_ptr(), n, out.data_ptr())
        return out


# ------------------------------------------------------------------------------------------- driver
class DistributedMdbg:
    def __init__(self, engine, comm, torch, profile=False):
        self.e, self.c, self.t = engine, comm, torch
        self.profile, self.times = profile, {}

    def _tick(self, name, t0):
        """accumulates wall time per stage when profiling (forces a device sync, so only for diagnosis)"""
        if not self.profile:
            return 0.0
        import time
        if self.t.cuda.is_available():
            self.t.cuda.synchronize()
        now = time.perf_counter()
        self.times[name] = self.times.get(name, 0.0) + (now - t0) * 1e3
        return now

    # process_read_aux over this rank's shard of a batch (src/main.rs:730-785), distributed
    def ingest_device(self, d_bases, d_offsets, n_reads, n_bases, first_ordinal):
        import time
        t0 = time.perf_counter()
        self.e.sketch_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal)
        self._tick("sketch", t0)
        self.exchange()

    def ingest_device_chunked(self, d_bases, offsets_dev, plan, first_ordinal):
        """Same as ingest_device for a batch cut into chunks of whole reads (plan from plan_chunks): a producer thread
        sketches chunk c+1 and packs its records while this thread exchanges and inserts chunk c.  Needs an engine with
        a separate table context; offsets_dev is the device offsets array as an int64 tensor."""
        import queue
        import threading
        t, e = self.t, self.e
        q = queue.Queue()
        sent = threading.Semaphore(1)          # route_out of the engine is reused: pack chunk c+1 only after chunk c was sent
        stop = threading.Event()               # consumer failed: the producer must not stay blocked on `sent`
        err = []

        def producer():
            try:
                for (r0, r1, a, nb) in plan:
                    offs_c = (offsets_dev[r0:r1 + 1] - a).contiguous()
                    t.cuda.synchronize()
                    e.sketch_device(e.base_ptr(d_bases, a), offs_c.data_ptr(), r1 - r0, nb, first_ordinal + r0)
                    while not sent.acquire(timeout=0.2):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    q.put(e.route_pack(self.c.world))
            except BaseException as ex:        # noqa: BLE001
                err.append(ex)
                q.put(None)

        th = threading.Thread(target=producer)
        th.start()
        try:
            for _ in plan:
                item = q.get()
                # a rank whose producer failed still takes part in the round (with nothing to send): the peers are inside the
                # same collective and would otherwise wait for it forever; the error is raised once the rounds are over
                recs, counts = item if item is not None else e.route_empty(self.c.world)
                recv, _ = self.c.alltoallv(recs, counts, getattr(e, "alloc_records", None))
                t.cuda.synchronize()
                sent.release()
                e.insert_records(recv)
        finally:
            stop.set()
            th.join()
        if err:
            raise err[0]

    def ingest_host(self, bases, offsets, first_ordinal):
        self.e.sketch_host(bases, offsets, first_ordinal)
        self.exchange()

    def exchange(self):
        import time
        t0 = time.perf_counter()
        recs, counts = self.e.route_pack(self.c.world)
        t0 = self._tick("route_pack", t0) or t0
        recv, _ = self.c.alltoallv(recs, counts, getattr(self.e, "alloc_records", None))
        t0 = self._tick("alltoall_records", t0) or t0
        self.e.insert_records(recv)
        self._tick("insert_records", t0)

    def finalize(self):
        """-> this rank's partition of the node table (dict of tensors) plus global counters; rows carry their global
        `row` (= position in index order), so concatenating all partitions and sorting by row gives the reference's table"""
        import time
        t, e, c = self.t, self.e, self.c
        t0 = time.perf_counter()
        all_ranges = c.allgather_obj(e.ranges)
        spans = [(min(fo for fo, _ in rl), max(fo + n for fo, n in rl)) if rl else None for rl in all_ranges]
        live = sorted((s[0], r) for r, s in enumerate(spans) if s)
        for (a, ra), (b, rb) in zip(live, live[1:]):
            assert spans[ra][1] <= b, "each rank must hold one contiguous range of read ordinals"
        ex = e.export(c.world, [lo for lo, _ in live], [r for _, r in live])       # both query lists, bucketed by answering rank
        t0 = self._tick("fin_export", t0) or t0
        # 1. first sightings -> DbgEntry.index and row, answered by the rank that sketched that read
        recv, rc = c.alltoallv(t.stack([ex["first"], ex["solid"]], 1), ex["counts_all"])
        t0 = self._tick("fin_query_first", t0) or t0
        rf, rs, tf, ts = e.resolve_first(recv[:, 0].contiguous(), recv[:, 1].contiguous())
        t0 = self._tick("fin_resolve_first", t0) or t0
        totals = c.allgather_obj((tf, ts))
        base_f = base_s = 0
        for _, r in live:
            if r == c.rank:
                break
            base_f += totals[r][0]
            base_s += totals[r][1]
        ans, _ = c.alltoallv(t.stack([rf + base_f, rs + base_s], 1), rc)          # comes back in list-A order
        t0 = self._tick("fin_reply_first", t0) or t0
        # 2. metadata of the A-th sighting, solid nodes only
        recv2, rc2 = c.alltoallv(ex["ath"], ex["counts_solid"])
        meta_local = e.resolve_meta(recv2.contiguous())
        t0 = self._tick("fin_query+resolve_meta", t0) or t0
        meta, _ = c.alltoallv(meta_local, rc2)                                     # list-S order
        t0 = self._tick("fin_reply_meta", t0) or t0
        keys = e.keys(ex["slot"])
        sel = ex["idx_all"]
        self._tick("fin_keys", t0)
        n_nodes = sum(x[1] for x in totals)
        n_before = sum(x[0] for x in totals)
        return dict(keys=keys, index=ans[sel, 0], row=ans[sel, 1], abundance=ex["count"] & 0xFFFF, seqlen=meta[:, 0] & 0xFFFFFFFF,
                    reversed=(meta[:, 0] >> 32) & 1, shift_full=meta[:, 1:3], src_read=meta[:, 3], src_start=meta[:, 4], src_end=meta[:, 5],
                    n_nodes=n_nodes, n_nodes_before=n_before, n_local=int(sel.shape[0]))

    def finalize_device_count(self):
        part = self.finalize()
        self.last_local = part["n_local"]
        return part["n_nodes"]


class ReplicatedMdbg:
    """Second multi-GPU mode (include/mdbg_hip.h "replicated sketches, partitioned table"): all-gather of the sketches,
    every rank windows the global sketch and inserts the k-min-mers it owns, one sum-all-reduce of two bitmaps at finalize.
    Communication per rank grows with the number of ranks (all-gather), so this is the intra-node mode; DistributedMdbg
    (all-to-all of k-min-mer records, volume per rank independent of the world size) is the one that scales out."""

    def __init__(self, engine, comm, torch):
        self.e, self.c, self.t = engine, comm, torch
        self.sized = False             # sketch store sized for the whole exchange (zero-copy receives must not move it)
        engine.set_partition(comm.world, comm.rank)

    def reset(self):
        self.e.reset()

    def ingest_device(self, d_bases, d_offsets, n_reads, n_bases, first_ordinal):
        """one collective round; a rank that has no reads for it passes n_reads = 0"""
        if n_reads:
            self.e.sketch_device(d_bases, d_offsets, n_reads, n_bases, first_ordinal)
        self._finish([self._share_begin(1, [], have_batch=n_reads > 0)])

    def ingest_host(self, bases, offsets, first_ordinal):
        n_reads = len(offsets) - 1
        if n_reads:
            self.e.sketch_host(bases, offsets, first_ordinal)
        self._finish([self._share_begin(1, [], have_batch=n_reads > 0)])

    def ingest_device_chunked(self, d_bases, offsets_dev, plan, first_ordinal):
        """One batch cut into chunks of whole reads (plan_chunks): while chunk c travels to the peers (RCCL send/recv
        pairs straight into their sketch stores) the tile kernel already works on chunk c+1; windows are inserted once
        everything has arrived.  offsets_dev: the device offsets array as an int64 tensor."""
        t, e = self.t, self.e
        pend = []
        for (r0, r1, a, nb) in plan:
            if r1 > r0:
                offs_c = (offsets_dev[r0:r1 + 1] - a).contiguous()
                if offs_c.is_cuda:
                    t.cuda.current_stream().synchronize()
                self._with_room(pend, lambda: e.sketch_device(e.base_ptr(d_bases, a), offs_c.data_ptr(), r1 - r0, nb, first_ordinal + r0))
            pend.append(self._share_begin(len(plan), pend, have_batch=r1 > r0))     # an empty chunk still takes part in the round
        self._finish(pend)

    def ingest_host_chunks(self, chunks):
        """[(bases, offsets, first_ordinal) or None] host batches, pipelined like ingest_device_chunked; every rank passes
        the same number of entries, None = no reads in that round"""
        pend = []
        for ch in chunks:
            have = ch is not None and len(ch[1]) > 1
            if have:
                bases, offsets, first = ch
                self._with_room(pend, lambda: self.e.sketch_host(bases, offsets, first))
            pend.append(self._share_begin(len(chunks), pend, have_batch=have))
        self._finish(pend)

    def _with_room(self, pend, fn):
        """run fn; if the sketch store would have to move while receives are in flight (MDBG_E_STATE), complete them first"""
        try:
            return fn()
        except Exception as ex:                # noqa: BLE001
            if getattr(ex, "code", None) != -6 or not pend:
                raise
        self._drain(pend)
        return fn()

    def _share_begin(self, n_chunks, pend, have_batch=True):
        """start sending the batch sketched last (nothing if this rank had no reads for the round) to every peer and
        receiving theirs -> pending item"""
        t, e, c = self.t, self.e, self.c
        if have_batch:
            h, p, off, first, n = e.last_sketch()
        else:
            dev = getattr(e, "device", None)
            h, p, off, first, n = (t.empty(0, dtype=t.int64, device=dev), t.empty(0, dtype=t.int32, device=dev),
                                   t.zeros(1, dtype=t.int64, device=dev), 0, 0)
        # every rank counts the windows of its own batch per owner once and ships the counts with the sizes, so that no
        # rank has to re-count a foreign sketch to size its table; engines that can, also list them per owner (4 bytes per
        # window travel with the sketch), so that no rank has to scan a foreign sketch for its windows either
        use_lists = hasattr(e, "owner_lists") and 1 < c.world <= 64
        counts, lists = None, None
        if have_batch and c.world > 1:
            if use_lists:
                counts, lists = e.owner_lists(c.world)
            elif hasattr(e, "owner_counts"):
                counts = e.owner_counts(c.world)
        meta = c.allgather_i64([h.shape[0], n, first, 0 if counts is None and n else 1] + (counts if counts is not None else [0] * c.world))
        peers = [r for r in range(c.world) if r != c.rank]
        if not peers:
            return None
        if not self.sized and hasattr(e, "store_reserve"):
            # size the store once, before anything is in flight: all chunks of all ranks, from the first chunk's counts
            self._drain(pend)
            e.store_reserve(int(sum(x[0] for x in meta) * n_chunks * 1.2) + (1 << 20), int(sum(x[1] for x in meta) * n_chunks * 1.2) + 4096 * n_chunks * c.world)
            self.sized = True
        bufs = self._with_room(pend, lambda: e.reserve_import([int(meta[r][0]) for r in peers]))
        if have_batch:
            h, p, off, first, n = e.last_sketch()          # views into the resident store: taken AFTER the reservation, which may move it
        offs = [t.empty(int(meta[r][1]) + 1, dtype=t.int64, device=off.device) for r in peers]
        sends = [(r, [h, p, off]) for r in peers]
        recvs = [(r, [bufs[i][0], bufs[i][1], offs[i]]) for i, r in enumerate(peers)]
        wl = [None] * len(peers)
        if use_lists:
            base = [0]
            for x in (counts if counts is not None else [0] * c.world):
                base.append(base[-1] + x)
            empty = t.empty(0, dtype=t.int32, device=off.device)
            for i, r in enumerate(peers):
                sends[i][1].append(lists[2 * base[r]:2 * base[r + 1]] if lists is not None else empty)
                wl[i] = t.empty(2 * int(meta[r][4 + c.rank]) if meta[r][3] else 0, dtype=t.int32, device=off.device)
                recvs[i][1].append(wl[i])
        handle = c.exchange(sends, recvs)
        return handle, [(bufs[i][2], offs[i], int(meta[r][2]), int(meta[r][4 + c.rank]) if meta[r][3] else None, wl[i] if (use_lists and meta[r][3]) else None)
                        for i, r in enumerate(peers)]

    def _drain(self, pend):
        for item in pend:
            if item is not None and item[0] is not None:
                item[0].wait()
                for token, off, first, owned, wlist in item[1]:
                    if wlist is not None:
                        self.e.commit_import(token, off, first, owned, wlist)
                    elif owned is None:
                        self.e.commit_import(token, off, first)
                    else:
                        self.e.commit_import(token, off, first, owned)
        pend[:] = []

    def _finish(self, pend):
        self._drain(pend)
        self.e.insert_owned()

    def finalize(self, counts_only=False):
        bf, bs = self.e.finalize_begin()
        self.c.allreduce_sum_(bf)      # every bit is set by exactly one rank (distinct keys have distinct first sightings)
        self.c.allreduce_sum_(bs)
        if bf.is_cuda:
            self.t.cuda.current_stream().synchronize()     # the engine works on its own stream
        return self.e.finalize_end(True) if counts_only else self.e.finalize_end()

    def finalize_device_count(self):
        """finalize with the node rows left on the device (what the benchmark times) -> global node count"""
        part = self.finalize(counts_only=True)
        self.last_local = part["n_local"]
        return part["n_nodes"]


def plan_chunks(offsets_host, n_chunks, keep_empty=False):
    """cut a batch (host copy of its offsets) into n_chunks runs of whole reads with roughly equal bases:
    -> [(r0, r1, aligned_byte, n_bytes)]: the chunk is reads [r0, r1), passed with base pointer + aligned_byte (a multiple
    of 64 <= offsets[r0]: 16-byte aligned in ASCII and in the 2-bit packed layout) and offsets rebased by it; n_bytes = offsets[r1] - aligned_byte.  keep_empty: always n_chunks
    entries (every rank of a collective driver must run the same number of rounds, however few reads it holds)"""
    import numpy as np
    o = np.asarray(offsets_host, dtype=np.uint64)
    n = len(o) - 1
    cuts = [0]
    for c in range(1, n_chunks):
        r = int(np.searchsorted(o, o[-1] * c // n_chunks))
        cuts.append(min(max(r, cuts[-1]), n))
    cuts.append(n)
    plan = []
    for r0, r1 in zip(cuts, cuts[1:]):
        if r1 > r0 or keep_empty:
            a = int(o[r0]) // 64 * 64
            plan.append((r0, r1, a, int(o[r1]) - a))
    return plan


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
