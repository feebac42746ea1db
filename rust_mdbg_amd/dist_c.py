"""ctypes mirror of include/mdbg_dist.h (the multi-GPU layer of libmdbg_hip.so that talks to RCCL itself) plus the few lines a
torch.distributed launch needs to hand it an ncclComm_t.  rust_mdbg_amd/dist.py is the older Python driver of the same protocol."""
import ctypes as C
import os

from . import api


class Comm(C.Structure):                 # mdbg_comm
    _fields_ = [("self", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("allgather_u64", C.c_void_p), ("exchange", C.c_void_p),
                ("allreduce_sum_u64", C.c_void_p), ("exchange_begin", C.c_void_p), ("exchange_wait", C.c_void_p)]


class UniqueId(C.Structure):             # ncclUniqueId, passed by value (rccl.h:187,220)
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    import torch
    L = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    return L


def rccl_comm(rank, world, dist=None):
    """a fresh ncclComm_t over the `world` processes of a torch.distributed job (rank 0's unique id travels through `dist`)"""
    L = _rccl()
    uid = UniqueId()
    if rank == 0 and L.ncclGetUniqueId(C.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    if world > 1:
        # all 128 bytes: the id is binary (bytes(uid.internal) on a c_char array stops at the first NUL byte)
        box = [C.string_at(C.byref(uid), C.sizeof(uid)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if len(box[0]) != C.sizeof(uid):
            raise RuntimeError("ncclUniqueId: %d bytes received, %d expected" % (len(box[0]), C.sizeof(uid)))
        C.memmove(C.byref(uid), box[0], C.sizeof(uid))
    comm = C.c_void_p()
    if L.ncclCommInitRank(C.byref(comm), world, uid, rank) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    return comm, L


class Xfer(C.Structure):            # mdbg_xfer
    _fields_ = [("peer", C.c_uint32), ("d_ptr", C.c_void_p), ("bytes", C.c_uint64)]


class HostStagedComm:
    """An mdbg_comm whose transfers are staged through HOST memory and carried by a torch.distributed process group of CPU tensors (gloo):
    device -> host copy, send / recv between the processes, host -> device copy.  It is the transport for DRY RUNS of the multi-process path
    where there is no GPU per rank (several ranks share one device, where RCCL refuses to start) — every line of the multi-GPU layer above
    the communicator runs exactly as under RCCL, the bytes just take the slow road.  NOT the RCCL path; `bench.py --comm host` labels its line so."""
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64))
    EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_uint32, C.POINTER(Xfer), C.c_uint32)
    AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)

    def __init__(self, dist, rank, world, group=None):
        import numpy as np
        import torch
        self.rank, self.world = rank, world
        hip = C.CDLL("libamdhip64.so")        # (the copy torch loaded: same SONAME)
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        D2H, H2D = 2, 1

        def allgather(_, send, n, recv):
            try:
                mine = torch.from_numpy(np.array([send[i] for i in range(n)], dtype=np.uint64).view(np.int64))
                out = [torch.zeros(n, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(out, mine, group=group)
                flat = np.concatenate([o.numpy() for o in out]).view(np.uint64) if n else np.zeros(0, np.uint64)
                for i in range(world * n):
                    recv[i] = int(flat[i])
                return 0
            except Exception:
                return -4

        def exchange(_, sends, ns, recvs, nr):
            try:
                ops, bufs, seq_s, seq_r = [], [], {}, {}
                for i in range(nr):                       # receives first: every peer's sends then find their match whatever the order of arrival
                    p, nbytes = int(recvs[i].peer), int(recvs[i].bytes)
                    t = torch.empty(nbytes, dtype=torch.uint8)
                    tag = seq_r.get(p, 0); seq_r[p] = tag + 1      # transfers between a pair are matched in the order they are listed
                    ops.append(dist.irecv(t, src=p, group=group, tag=tag))
                    bufs.append((t, recvs[i].d_ptr, nbytes))
                for i in range(ns):
                    p, nbytes = int(sends[i].peer), int(sends[i].bytes)
                    t = torch.empty(nbytes, dtype=torch.uint8)
                    if hip.hipMemcpy(t.data_ptr(), sends[i].d_ptr, nbytes, D2H) != 0:
                        return -4
                    tag = seq_s.get(p, 0); seq_s[p] = tag + 1
                    ops.append(dist.isend(t, dst=p, group=group, tag=tag))
                for o in ops:
                    o.wait()
                for t, d_ptr, nbytes in bufs:
                    if hip.hipMemcpy(d_ptr, t.data_ptr(), nbytes, H2D) != 0:
                        return -4
                return 0
            except Exception:
                return -4

        def allreduce(_, d_buf, n):
            try:
                a = torch.zeros(int(n), dtype=torch.int64)
                if n and hip.hipMemcpy(a.data_ptr(), d_buf, n * 8, D2H) != 0:
                    return -4
                dist.all_reduce(a, group=group)            # (two's complement: the sum of int64 is the sum of u64 modulo 2^64)
                if n and hip.hipMemcpy(d_buf, a.data_ptr(), n * 8, H2D) != 0:
                    return -4
                return 0
            except Exception:
                return -4

        self.fns = (self.AG(allgather), self.EX(exchange), self.AR(allreduce))      # kept alive as long as the mdbg_dist
        cm = Comm()
        cm.self = None
        cm.rank, cm.world = rank, world
        cm.allgather_u64 = C.cast(self.fns[0], C.c_void_p)
        cm.exchange = C.cast(self.fns[1], C.c_void_p)
        cm.allreduce_sum_u64 = C.cast(self.fns[2], C.c_void_p)
        self.table = cm


class DistMdbg:
    """mdbg_dist: one per process / GPU.  transport "rccl" (default): the library's own RCCL calls; "host": HostStagedComm over `dist` (dry runs)"""

    def __init__(self, k, l, density, min_abundance, rank, world, dist=None, device=-1, reads_already_hpc=False, transport="rccl"):
        self.L = api.load_library()
        L = self.L
        L.mdbg_comm_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Comm)]
        L.mdbg_dist_create.restype = C.c_void_p
        L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(Comm), C.POINTER(C.c_int)]
        L.mdbg_dist_ctx.restype = C.c_void_p
        L.mdbg_dist_ctx.argtypes = [C.c_void_p]
        L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.mdbg_dist_ingest_batch_packed_device.argtypes = [C.c_void_p, C.POINTER(api.PackedBatch), C.c_uint64, C.c_uint64]
        L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.mdbg_dist_reset.argtypes = [C.c_void_p, C.c_uint32]
        L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
        L.mdbg_dist_set_exchange.argtypes = [C.c_void_p, C.c_uint32]
        L.mdbg_dist_destroy.argtypes = [C.c_void_p]
        L.mdbg_dist_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mdbg_dist_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint64), C.c_int]
        L.mdbg_dist_stage_name.argtypes = [C.c_uint32]; L.mdbg_dist_stage_name.restype = C.c_char_p
        self.comm = self.rccl = self.host_comm = None
        if transport == "host":
            self.host_comm = HostStagedComm(dist, rank, world)
            vt = self.host_comm.table
        else:
            self.comm, self.rccl = rccl_comm(rank, world, dist)
            vt = Comm()
            self._chk(L.mdbg_comm_rccl(self.comm, rank, world, C.byref(vt)))
        P = api.Params(k=k, l=l, density=density, min_abundance=min_abundance, reads_already_hpc=int(reads_already_hpc), device=device, flags=0,
                       table_capacity_hint=0)
        err = C.c_int()
        self.h = L.mdbg_dist_create(C.byref(P), C.byref(vt), C.byref(err))
        if not self.h:
            raise api.MdbgError(err.value, L.mdbg_strerror(err.value).decode())
        self.k = k

    def _chk(self, e):
        if e:
            raise api.MdbgError(e, self.L.mdbg_strerror(e).decode())

    def set_pipeline(self, chunks):
        """cut every ingest call into `chunks` rounds whose exchange overlaps the next chunk's sketch kernel (same value on every rank)"""
        self._chk(self.L.mdbg_dist_set_pipeline(self.h, chunks))

    def set_exchange(self, whole):
        """whole = False (default): per peer its window list and only the hashes those windows need; True: every sketch entire to every rank
        (needed for reset(new_k) on the resident sketches)"""
        self._chk(self.L.mdbg_dist_set_exchange(self.h, 1 if whole else 0))

    def ingest_device(self, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal):
        self._chk(self.L.mdbg_dist_ingest_batch_device(self.h, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal))

    def ingest_packed_device(self, d_words, d_offsets, n_reads, n_bases, first_read_ordinal):
        b = api.PackedBatch(d_words, d_offsets, n_reads, 0, 0, 0)
        self._chk(self.L.mdbg_dist_ingest_batch_packed_device(self.h, C.byref(b), n_bases, first_read_ordinal))

    def finalize(self):
        """-> (mdbg_nodes with DEVICE pointers: this rank's partition, device pointer to the global rows, global node count)"""
        nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
        self._chk(self.L.mdbg_dist_finalize(self.h, C.byref(nd), C.byref(row), C.byref(ng)))
        return nd, row.value, int(ng.value)

    def set_timing(self, level):
        """mdbg_set_timing on the layer's context (0: no HIP events, 1: tile kernel only, 2: stages too)"""
        self._chk(self.L.mdbg_set_timing(C.c_void_p(self.L.mdbg_dist_ctx(self.h)), level))

    def nodes_digest(self, nd):
        """(sum, xor) over this rank's partition (a table of finalize()): the ranks' digests add / XOR up to the one-GPU table's (include/mdbg_hip.h, mdbg_nodes_digest)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.mdbg_nodes_digest(C.c_void_p(self.L.mdbg_dist_ctx(self.h)), C.byref(nd), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def reset(self, new_k=0):
        self._chk(self.L.mdbg_dist_reset(self.h, new_k))

    def traffic(self):
        """-> (bytes received, bytes sent, position queries sent) by this rank since create / reset(0)"""
        a, b, q = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.mdbg_dist_traffic(self.h, C.byref(a), C.byref(b), C.byref(q)))
        return int(a.value), int(b.value), int(q.value)

    def stage_ms(self, reset=False):
        """-> ({stage: host ms of this rank since create / the last reset}, rounds)"""
        out = (C.c_double * 32)(); nr = C.c_uint64()
        self._chk(self.L.mdbg_dist_stage_ms(self.h, out, 32, C.byref(nr), 1 if reset else 0))
        d = {}
        for i in range(32):
            nm = self.L.mdbg_dist_stage_name(i)
            if not nm:
                break
            d[nm.decode()] = float(out[i])
        return d, int(nr.value)

    def close(self):
        if self.h:
            self.L.mdbg_dist_destroy(self.h)
            self.h = None
            if self.rccl is not None:
                self.rccl.ncclCommDestroy(self.comm)
