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


class DistMdbg:
    """mdbg_dist over RCCL: one per process / GPU"""

    def __init__(self, k, l, density, min_abundance, rank, world, dist=None, device=-1, reads_already_hpc=False):
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

    def reset(self, new_k=0):
        self._chk(self.L.mdbg_dist_reset(self.h, new_k))

    def traffic(self):
        """-> (bytes received, bytes sent, position queries sent) by this rank since create / reset(0)"""
        a, b, q = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.mdbg_dist_traffic(self.h, C.byref(a), C.byref(b), C.byref(q)))
        return int(a.value), int(b.value), int(q.value)

    def close(self):
        if self.h:
            self.L.mdbg_dist_destroy(self.h)
            self.h = None
            self.rccl.ncclCommDestroy(self.comm)
