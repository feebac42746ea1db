"""ctypes binding of include/mdbg_hip.h — the host-side mirror of rust-mdbg's per-read path.

Names follow the reference: `Params` (src/main.rs:92-114), `Mdbg.ingest` = process_read_aux over a batch
(src/main.rs:730-785), `Mdbg.sketch` = Read::extract (src/read.rs:85-90), `Mdbg.finalize` = the abundance filter
plus the read-only node view the graph emitter walks (src/main.rs:922-929,1014-1016).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MDBG_OK, MDBG_E_PARAM, MDBG_E_ALPHABET, MDBG_E_CAPACITY, MDBG_E_DEVICE, MDBG_E_NOMEM, MDBG_E_STATE, MDBG_E_IO = 0, -1, -2, -3, -4, -5, -6, -7
FLAG_FORCE_GENERIC = 1   # mdbg_params.flags bit 0: every tile takes the generic exact kernel (testing)


class MdbgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mdbg error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("l", C.c_uint32), ("density", C.c_double), ("min_abundance", C.c_uint32),
                ("reads_already_hpc", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("table_capacity_hint", C.c_uint64), ("scheme", C.c_uint32), ("syncmer_s", C.c_uint32), ("reserved", C.c_uint64 * 3)]


class Nodes(C.Structure):
    _fields_ = [("n", C.c_uint64), ("k", C.c_uint32), ("keys", C.POINTER(C.c_uint64)), ("index", C.POINTER(C.c_uint32)),
                ("abundance", C.POINTER(C.c_uint16)), ("seqlen", C.POINTER(C.c_uint32)), ("shift", C.POINTER(C.c_uint16)),
                ("shift_full", C.POINTER(C.c_uint64)), ("src_read", C.POINTER(C.c_uint64)), ("src_start", C.POINTER(C.c_uint64)),
                ("src_end", C.POINTER(C.c_uint64)), ("reversed", C.POINTER(C.c_uint8)), ("n_distinct", C.c_uint64),
                ("n_wrapped", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_minimizers", C.c_uint64), ("n_windows", C.c_uint64),
                ("n_distinct", C.c_uint64), ("table_capacity", C.c_uint64), ("n_slow_tiles", C.c_uint64), ("n_tiles", C.c_uint64),
                ("ms_sketch", C.c_double), ("ms_insert", C.c_double), ("ms_finalize", C.c_double), ("ms_sketch_tile", C.c_double),
                ("n_sketch_tile_launches", C.c_uint64), ("n_sketch_tile_bases", C.c_uint64), ("tile_bases", C.c_uint64),
                ("n_link_matches", C.c_uint64), ("reserved", C.c_uint64 * 3)]


class RoutedLists(C.Structure):
    _fields_ = [("n_all", C.c_uint64), ("n_solid", C.c_uint64), ("d_first", C.c_void_p), ("d_solid", C.c_void_p), ("d_ath", C.c_void_p),
                ("d_count", C.c_void_p), ("d_slot", C.c_void_p), ("d_idx_all", C.c_void_p), ("counts_all", C.c_uint64 * 64),
                ("counts_solid", C.c_uint64 * 64)]


class SketchStore(C.Structure):
    _fields_ = [("n_minimizers", C.c_uint64), ("n_reads", C.c_uint64), ("d_hashes", C.c_void_p), ("d_positions", C.c_void_p),
                ("d_read_offsets", C.c_void_p)]


class BatchInfo(C.Structure):
    _fields_ = [("store_offset", C.c_uint64), ("n_minimizers", C.c_uint64), ("first_slot", C.c_uint64), ("n_reads", C.c_uint64),
                ("first_read_ordinal", C.c_uint64), ("d_hashes", C.c_void_p), ("d_positions", C.c_void_p), ("d_read_offsets", C.c_void_p)]


class EdgeList(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n1", C.POINTER(C.c_uint32)), ("o1", C.POINTER(C.c_uint8)), ("n2", C.POINTER(C.c_uint32)),
                ("o2", C.POINTER(C.c_uint8)), ("overlap", C.POINTER(C.c_uint32)), ("presimp_removed", C.c_uint64)]


class PackedBatch(C.Structure):          # mdbg_packed_batch
    _fields_ = [("words", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64), ("exc_pos", C.c_void_p),
                ("exc_val", C.c_void_p), ("n_exc", C.c_uint64)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("n_reads", C.c_uint64), ("mean_len", C.c_uint32),
                ("sd_len", C.c_uint32), ("min_len", C.c_uint32), ("max_len", C.c_uint32), ("err_ppm", C.c_uint32),
                ("reserved", C.c_uint32)]


EXPORTS = ["mdbg_abi_version", "mdbg_build_flags", "mdbg_create", "mdbg_destroy", "mdbg_finalize_device", "mdbg_finalize_gfa", "mdbg_nodes_digest", "mdbg_set_timing", "mdbg_ingest_batch", "mdbg_ingest_batch_device", "mdbg_sketch_only",
           "mdbg_finalize", "mdbg_reset", "mdbg_get_stats", "mdbg_strerror", "mdbg_last_error", "mdbg_sketch_device",
           "mdbg_insert_resident", "mdbg_route_pack", "mdbg_insert_records", "mdbg_sync", "mdbg_synth_reads_device", "mdbg_copy_to_host", "mdbg_copy_to_device",
           "mdbg_routed_export", "mdbg_resolve_first", "mdbg_resolve_meta", "mdbg_routed_keys", "mdbg_arena_reserve",
           "mdbg_set_partition", "mdbg_sketch_view", "mdbg_ingest_sketch", "mdbg_finalize_begin", "mdbg_finalize_end",
           "mdbg_store_reserve", "mdbg_sketch_reserve", "mdbg_sketch_commit", "mdbg_last_batch", "mdbg_owner_counts", "mdbg_graph_edges", "mdbg_graph_edges_device",
           "mdbg_ingest_batch_packed", "mdbg_ingest_batch_packed_device", "mdbg_sketch_packed_device", "mdbg_pack_device", "mdbg_query_batch", "mdbg_owner_lists", "mdbg_sketch_commit_listed", "mdbg_mark", "mdbg_rewind", "mdbg_set_lmer_filter",
           "mdbg_release_cached_memory", "mdbg_host_alloc", "mdbg_host_free", "mdbg_host_is_pinned", "mdbg_dbg_segments_ms"]


def lib_path():
    return os.path.join(_HERE, "libmdbg_hip.so")


def load_library():
    """Loads libmdbg_hip.so.  No fallback: a missing library is an error (build with __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("libmdbg_hip.so not built (%s); run `make -C rust_mdbg_amd/csrc`" % p)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library (linked against
    # /opt/rocm's) is loaded first, a later `import torch` finds no GPU.  Loading torch first makes both share torch's
    # copy (same SONAME).  Plumbing only - nothing of torch is used here.  MDBG_NO_TORCH_PRELOAD=1 disables it.
    import sys
    if "torch" not in sys.modules and not os.environ.get("MDBG_NO_TORCH_PRELOAD"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    L = C.CDLL(p)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.mdbg_abi_version.restype = u32
    L.mdbg_build_flags.restype = u32
    L.mdbg_release_cached_memory.restype = u64
    L.mdbg_host_alloc.restype = vp
    L.mdbg_host_alloc.argtypes = [C.c_size_t]
    L.mdbg_host_free.restype = None
    L.mdbg_host_free.argtypes = [vp]
    L.mdbg_host_is_pinned.argtypes = [vp]
    L.mdbg_create.restype = vp
    L.mdbg_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_int)]
    L.mdbg_destroy.restype = None
    L.mdbg_destroy.argtypes = [vp]
    L.mdbg_ingest_batch.argtypes = [vp, vp, vp, u64, u64]
    L.mdbg_ingest_batch_device.argtypes = [vp, vp, vp, u64, u64, u64]
    L.mdbg_sketch_device.argtypes = [vp, vp, vp, u64, u64, u64]
    L.mdbg_insert_resident.argtypes = [vp]
    L.mdbg_ingest_batch_packed.argtypes = [vp, C.POINTER(PackedBatch), u64]
    L.mdbg_ingest_batch_packed_device.argtypes = [vp, C.POINTER(PackedBatch), u64, u64]
    L.mdbg_sketch_packed_device.argtypes = [vp, C.POINTER(PackedBatch), u64, u64]
    L.mdbg_pack_device.argtypes = [vp, vp, u64, vp, vp, vp, u64, C.POINTER(u64)]
    L.mdbg_query_batch.argtypes = [vp, vp, vp, u64, C.POINTER(C.POINTER(u32)), C.POINTER(C.POINTER(u64)), C.POINTER(u64)]
    L.mdbg_query_batch.restype = C.c_int
    L.mdbg_sketch_only.argtypes = [vp, vp, vp, u64, C.POINTER(C.POINTER(u64)), C.POINTER(C.POINTER(u64)), C.POINTER(C.POINTER(u64)), C.POINTER(u64)]
    L.mdbg_finalize.argtypes = [vp, C.POINTER(Nodes)]
    L.mdbg_finalize_device.argtypes = [vp, C.POINTER(Nodes)]
    L.mdbg_finalize_gfa.argtypes = [vp, C.POINTER(Nodes)]
    L.mdbg_set_timing.argtypes = [vp, u32]
    L.mdbg_nodes_digest.argtypes = [vp, C.POINTER(Nodes), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mdbg_reset.argtypes = [vp, u32]
    L.mdbg_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mdbg_strerror.restype = C.c_char_p
    L.mdbg_strerror.argtypes = [C.c_int]
    L.mdbg_last_error.restype = C.c_char_p
    L.mdbg_last_error.argtypes = [vp]
    L.mdbg_route_pack.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
    L.mdbg_routed_export.argtypes = [vp, u32, vp, vp, u32, C.POINTER(RoutedLists)]
    L.mdbg_resolve_first.argtypes = [vp, vp, vp, u64, vp, vp, C.POINTER(u64), C.POINTER(u64)]
    L.mdbg_resolve_meta.argtypes = [vp, vp, u64, vp]
    L.mdbg_routed_keys.argtypes = [vp, vp, u64, vp]
    L.mdbg_arena_reserve.argtypes = [vp, u64, C.POINTER(vp)]
    L.mdbg_set_partition.argtypes = [vp, u32, u32]
    L.mdbg_sketch_view.argtypes = [vp, C.POINTER(SketchStore)]
    L.mdbg_ingest_sketch.argtypes = [vp, vp, vp, vp, u64, u64]
    L.mdbg_store_reserve.argtypes = [vp, u64, u64]
    L.mdbg_sketch_reserve.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.mdbg_sketch_commit.argtypes = [vp, u64, u64, vp, u64, u64, u64]
    L.mdbg_owner_counts.argtypes = [vp, u32, vp]
    L.mdbg_owner_lists.argtypes = [vp, u32, vp, C.POINTER(vp)]
    L.mdbg_owner_lists.restype = C.c_int
    L.mdbg_sketch_commit_listed.argtypes = [vp, u64, u64, vp, u64, u64, vp, u64]
    L.mdbg_sketch_commit_listed.restype = C.c_int
    L.mdbg_set_lmer_filter.argtypes = [vp, vp, u64]
    L.mdbg_set_lmer_filter.restype = C.c_int
    L.mdbg_mark.argtypes = [vp, C.POINTER(u64)]
    L.mdbg_mark.restype = C.c_int
    L.mdbg_rewind.argtypes = [vp, u64]
    L.mdbg_rewind.restype = C.c_int
    L.mdbg_last_batch.argtypes = [vp, C.POINTER(BatchInfo)]
    L.mdbg_graph_edges.argtypes = [vp, C.c_float, C.POINTER(EdgeList)]
    L.mdbg_graph_edges_device.argtypes = [vp, C.c_float, C.POINTER(EdgeList)]
    L.mdbg_finalize_begin.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.mdbg_finalize_end.argtypes = [vp, C.POINTER(Nodes), C.POINTER(vp), C.POINTER(u64)]
    L.mdbg_insert_records.argtypes = [vp, vp, u64]
    L.mdbg_sync.argtypes = [vp]
    L.mdbg_copy_to_host.argtypes = [vp, vp, vp, u64]
    L.mdbg_copy_to_device.argtypes = [vp, vp, vp, u64]
    L.mdbg_synth_reads_device.argtypes = [vp, C.POINTER(SynthParams), u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    for f in ("mdbg_ingest_batch", "mdbg_ingest_batch_device", "mdbg_sketch_device", "mdbg_insert_resident", "mdbg_sketch_only",
              "mdbg_finalize", "mdbg_finalize_device", "mdbg_reset", "mdbg_get_stats", "mdbg_route_pack", "mdbg_insert_records", "mdbg_sync",
              "mdbg_synth_reads_device", "mdbg_copy_to_host", "mdbg_copy_to_device", "mdbg_routed_export", "mdbg_resolve_first",
              "mdbg_resolve_meta", "mdbg_routed_keys", "mdbg_arena_reserve", "mdbg_set_partition", "mdbg_sketch_view",
              "mdbg_ingest_sketch", "mdbg_finalize_begin", "mdbg_finalize_end", "mdbg_ingest_batch_packed",
              "mdbg_ingest_batch_packed_device", "mdbg_sketch_packed_device", "mdbg_pack_device"):
        getattr(L, f).restype = C.c_int
    _LIB = L
    return L


def release_cached_memory():
    """hands the device blocks the library keeps for reuse back to the runtime; returns the number of bytes released (include/mdbg_hip.h)"""
    return int(load_library().mdbg_release_cached_memory())


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def concat_reads(reads):
    """list[bytes] -> (uint8 bases, uint64 offsets[n+1]) — the batch layout of mdbg_ingest_batch"""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offs


class Mdbg:
    """One context = one device-resident sketch store + k-min-mer table (dbg_nodes, src/main.rs:595)."""

    def __init__(self, k, l, density, min_abundance=2, reads_already_hpc=False, device=-1, flags=0, table_capacity_hint=0, syncmer_s=None):
        self.L = load_library()
        self.params = Params(k=k, l=l, density=density, min_abundance=min_abundance, reads_already_hpc=int(reads_already_hpc),
                             device=device, flags=flags, table_capacity_hint=table_capacity_hint,
                             scheme=0 if syncmer_s is None else 1, syncmer_s=0 if syncmer_s is None else syncmer_s)      # syncmer_s: --syncmers -s
        err = C.c_int(0)
        self.h = self.L.mdbg_create(C.byref(self.params), C.byref(err))
        if not self.h:
            raise MdbgError(err.value, self.L.mdbg_strerror(err.value).decode())
        self.k = k

    def _chk(self, e):
        if e != 0:
            raise MdbgError(e, (self.L.mdbg_last_error(self.h) or self.L.mdbg_strerror(e)).decode())

    # --- process_read_aux over a batch (host buffers) ---
    def ingest(self, bases, offsets, first_read_ordinal=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self.L.mdbg_ingest_batch(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, first_read_ordinal))

    def ingest_reads(self, reads, first_read_ordinal=0):
        b, o = concat_reads(reads)
        self.ingest(b, o, first_read_ordinal)

    # --- device-resident buffers (raw pointers, e.g. torch.Tensor.data_ptr()) ---
    def ingest_device(self, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal=0):
        self._chk(self.L.mdbg_ingest_batch_device(self.h, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal))

    def sketch_device(self, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal=0):
        self._chk(self.L.mdbg_sketch_device(self.h, d_bases, d_offsets, n_reads, n_bases, first_read_ordinal))

    def insert_resident(self):
        self._chk(self.L.mdbg_insert_resident(self.h))

    # --- 2-bit packed input (mdbg_packed_batch) ---
    def ingest_packed(self, packed, first_read_ordinal=0):
        """packed: dict from rust_mdbg_amd.emit.pack_reads (host arrays words / offsets / exc_pos / exc_val)"""
        b = PackedBatch(packed["words"].ctypes.data, packed["offsets"].ctypes.data, len(packed["offsets"]) - 1,
                        packed["exc_pos"].ctypes.data, packed["exc_val"].ctypes.data, len(packed["exc_pos"]))
        self._chk(self.L.mdbg_ingest_batch_packed(self.h, C.byref(b), first_read_ordinal))

    def ingest_packed_device(self, d_words, d_offsets, n_reads, n_bases, first_read_ordinal=0, d_exc_pos=0, d_exc_val=0, n_exc=0, sketch_only=False):
        b = PackedBatch(d_words, d_offsets, n_reads, d_exc_pos, d_exc_val, n_exc)
        fn = self.L.mdbg_sketch_packed_device if sketch_only else self.L.mdbg_ingest_batch_packed_device
        self._chk(fn(self.h, C.byref(b), n_bases, first_read_ordinal))

    def pack_device(self, d_bases, n_bases, d_words, d_exc_pos=0, d_exc_val=0, exc_cap=0):
        n = C.c_uint64()
        self._chk(self.L.mdbg_pack_device(self.h, d_bases, n_bases, d_words, d_exc_pos, d_exc_val, exc_cap, C.byref(n)))
        return n.value

    def query(self, bases, offsets):
        """--read_stats: (counts u32[n_windows], per-read offsets u64[n_reads + 1])"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        pc, po, nw = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.c_uint64()
        self._chk(self.L.mdbg_query_batch(self.h, bases.ctypes.data, offsets.ctypes.data, n, C.byref(pc), C.byref(po), C.byref(nw)))
        return _np(pc, nw.value, np.uint32), _np(po, n + 1, np.uint64)

    def store_sketch(self):
        """host copy of the whole resident sketch store: hashes, positions, per-read offsets"""
        v = self.sketch_view()
        m, n = int(v.n_minimizers), int(v.n_reads)
        return dict(hashes=self.to_host(v.d_hashes, m * 8, np.uint64) if m else np.zeros(0, np.uint64),
                    pos=self.to_host(v.d_positions, m * 4, np.uint32).astype(np.uint64) if m else np.zeros(0, np.uint64),
                    off=self.to_host(v.d_read_offsets, (n + 1) * 8, np.uint64))

    # --- Read::extract seam ---
    def sketch(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        ph, pp, po = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        m = C.c_uint64()
        self._chk(self.L.mdbg_sketch_only(self.h, bases.ctypes.data, offsets.ctypes.data, n, C.byref(ph), C.byref(pp), C.byref(po), C.byref(m)))
        return dict(hashes=_np(ph, m.value, np.uint64), pos=_np(pp, m.value, np.uint64), off=_np(po, n + 1, np.uint64))

    def finalize(self, gfa_only=False):
        """-> the node table as numpy arrays.  gfa_only: the table stays on the device (where mdbg_graph_edges reads it) and only what the S lines of a .gfa
        print (index, length, abundance: 10 bytes per node instead of 8 k + 58) is copied to the host; the other entries are None"""
        nd = Nodes()
        if gfa_only:
            self._chk(self.L.mdbg_finalize_gfa(self.h, C.byref(nd)))
            n = int(nd.n)
            return dict(n_nodes=n, n_nodes_before=int(nd.n_distinct), n_wrapped=int(nd.n_wrapped), k=int(nd.k), keys=None, index=_np(nd.index, n, np.uint32),
                        abundance=_np(nd.abundance, n, np.uint16), seqlen=_np(nd.seqlen, n, np.uint32), shift=None, shift_full=None, src_read=None, src_start=None,
                        src_end=None, reversed=None)
        self._chk(self.L.mdbg_finalize(self.h, C.byref(nd)))
        n, k = nd.n, nd.k
        return dict(n_nodes=int(n), n_nodes_before=int(nd.n_distinct), n_wrapped=int(nd.n_wrapped),
                    keys=_np(nd.keys, n * k, np.uint64).reshape(n, k), index=_np(nd.index, n, np.uint32),
                    abundance=_np(nd.abundance, n, np.uint16), seqlen=_np(nd.seqlen, n, np.uint32),
                    shift=_np(nd.shift, 2 * n, np.uint16).reshape(n, 2), shift_full=_np(nd.shift_full, 2 * n, np.uint64).reshape(n, 2),
                    src_read=_np(nd.src_read, n, np.uint64), src_start=_np(nd.src_start, n, np.uint64),
                    src_end=_np(nd.src_end, n, np.uint64), reversed=_np(nd.reversed, n, np.uint8))

    def finalize_device(self):
        """node table left in device memory; returns the raw mdbg_nodes struct (device pointers)"""
        nd = Nodes()
        self._chk(self.L.mdbg_finalize_device(self.h, C.byref(nd)))
        return nd

    def set_timing(self, level):
        """0: no HIP events, 1: around the tile kernel only, 2 (default): also around the stages (include/mdbg_hip.h, mdbg_set_timing)"""
        self._chk(self.L.mdbg_set_timing(self.h, level))

    def nodes_digest(self, nd):
        """(sum, xor) over the rows of a DEVICE node table (finalize_device / DistMdbg.finalize): the order-free digest of {(key, abundance)}, include/mdbg_hip.h"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.mdbg_nodes_digest(self.h, C.byref(nd), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def reset(self, new_k):
        self._chk(self.L.mdbg_reset(self.h, new_k))
        if new_k:
            self.k = new_k

    def set_lmer_filter(self, codes):
        """--lmer-counts: codes = uint64 array of the selected l-mers' 2-bit codes, both orientations (emit.lmer_filter_from_counts);
        None switches the filter off.  Before the first batch."""
        if codes is None:
            self._chk(self.L.mdbg_set_lmer_filter(self.h, None, 0))
            return
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        keep = codes if len(codes) else np.zeros(1, dtype=np.uint64)        # a non-null pointer also for the empty selection
        self._chk(self.L.mdbg_set_lmer_filter(self.h, keep.ctypes.data, len(codes)))

    def mark(self):
        """-> token naming what is resident now (see rewind)"""
        mk = C.c_uint64()
        self._chk(self.L.mdbg_mark(self.h, C.byref(mk)))
        return int(mk.value)

    def rewind(self, mark):
        """forget every batch ingested after mark() and clear the node table; follow with reset(k) to re-window what is left"""
        self._chk(self.L.mdbg_rewind(self.h, mark))

    # --- multi-GPU stage calls (raw device pointers; see rust_mdbg_amd/dist.py for the driver) ---
    def route_pack(self, world):
        """-> (device pointer to bucketed records of k+2 u64, [records per destination])"""
        ptr = C.c_void_p()
        counts = (C.c_uint64 * 64)()
        self._chk(self.L.mdbg_route_pack(self.h, world, C.byref(ptr), counts))
        return ptr.value or 0, [int(counts[i]) for i in range(world)]

    def arena_reserve(self, n):
        """-> device pointer where n more routed records (k+2 u64 each) may be received in place"""
        ptr = C.c_void_p()
        self._chk(self.L.mdbg_arena_reserve(self.h, n, C.byref(ptr)))
        return ptr.value or 0

    def insert_records(self, d_records, n):
        self._chk(self.L.mdbg_insert_records(self.h, d_records, n))

    def routed_export(self, world, span_lo, span_rank):
        """owner side: the two query lists bucketed by answering rank -> RoutedLists (device pointers + host counts)"""
        lo = np.ascontiguousarray(span_lo, dtype=np.uint64)
        rk = np.ascontiguousarray(span_rank, dtype=np.uint32)
        out = RoutedLists()
        self._chk(self.L.mdbg_routed_export(self.h, world, lo.ctypes.data, rk.ctypes.data, len(lo), C.byref(out)))
        return out

    def resolve_first(self, d_ord, d_solid, n, d_rank_first, d_rank_solid):
        tf, ts = C.c_uint64(), C.c_uint64()
        self._chk(self.L.mdbg_resolve_first(self.h, d_ord, d_solid, n, d_rank_first, d_rank_solid, C.byref(tf), C.byref(ts)))
        return tf.value, ts.value

    def resolve_meta(self, d_ord, n, d_meta):
        self._chk(self.L.mdbg_resolve_meta(self.h, d_ord, n, d_meta))

    def routed_keys(self, d_slot, n, d_keys):
        self._chk(self.L.mdbg_routed_keys(self.h, d_slot, n, d_keys))

    # --- replicated-sketch multi-GPU mode ---
    def set_partition(self, world, rank):
        self._chk(self.L.mdbg_set_partition(self.h, world, rank))

    def sketch_view(self):
        s = SketchStore()
        self._chk(self.L.mdbg_sketch_view(self.h, C.byref(s)))
        return s

    def ingest_sketch(self, d_hashes, d_positions, d_read_offsets, n_reads, first_read_ordinal):
        self._chk(self.L.mdbg_ingest_sketch(self.h, d_hashes, d_positions, d_read_offsets, n_reads, first_read_ordinal))

    def graph_edges(self, presimp=0.01, raw=False):
        """edges of the node table of the last finalize(), built on the GPU (src/main.rs:1017-1117)
        -> dict(n1, o1, n2, o2, overlap, presimp_removed) of numpy arrays, o1/o2 as ASCII '+' / '-'; raw=True: the C struct"""
        e = EdgeList()
        self._chk(self.L.mdbg_graph_edges(self.h, presimp, C.byref(e)))
        if raw:
            return e
        n = int(e.n)
        g = lambda p, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t, copy=True) if n else np.zeros(0, t)
        return dict(n1=g(e.n1, np.uint32), o1=g(e.o1, np.uint8), n2=g(e.n2, np.uint32), o2=g(e.o2, np.uint8), overlap=g(e.overlap, np.uint32),
                    presimp_removed=int(e.presimp_removed))

    def graph_edges_device(self, presimp=0.01):
        """-> EdgeList with DEVICE pointers (valid until the next edge call)"""
        e = EdgeList()
        self._chk(self.L.mdbg_graph_edges_device(self.h, presimp, C.byref(e)))
        return e

    def store_reserve(self, n_minimizers_total, n_reads_total):
        self._chk(self.L.mdbg_store_reserve(self.h, n_minimizers_total, n_reads_total))

    def sketch_reserve(self, n_minimizers):
        """-> (d_hashes, d_positions, region): an uncommitted region at the end of the resident store to receive into"""
        h, p, r = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self.L.mdbg_sketch_reserve(self.h, n_minimizers, C.byref(h), C.byref(p), C.byref(r)))
        return h.value or 0, p.value or 0, r.value

    def sketch_commit(self, region, n_minimizers, d_read_offsets, n_reads, first_read_ordinal, owned_windows=None):
        """owned_windows: number of windows of this batch the context owns (from the sender's owner_counts), None = unknown"""
        ow = 0xFFFFFFFFFFFFFFFF if owned_windows is None else int(owned_windows)
        self._chk(self.L.mdbg_sketch_commit(self.h, region, n_minimizers, d_read_offsets, n_reads, first_read_ordinal, ow))

    def owner_lists(self, world):
        """-> (counts per owning rank, DEVICE pointer to the window lists bucketed by owner; an entry = two uint32: window start, read) for the
        batch registered last"""
        out = (C.c_uint64 * world)()
        p = C.c_void_p()
        self._chk(self.L.mdbg_owner_lists(self.h, world, out, C.byref(p)))
        return [int(x) for x in out], (p.value or 0)

    def sketch_commit_listed(self, region, n_minimizers, d_read_offsets, n_reads, first_read_ordinal, d_list, n_list):
        self._chk(self.L.mdbg_sketch_commit_listed(self.h, region, n_minimizers, d_read_offsets, n_reads, first_read_ordinal, d_list, n_list))

    def owner_counts(self, world):
        """windows of the batch registered last per owning rank -> list of `world` ints"""
        out = (C.c_uint64 * world)()
        self._chk(self.L.mdbg_owner_counts(self.h, world, out))
        return [int(x) for x in out]

    def last_batch(self):
        b = BatchInfo()
        self._chk(self.L.mdbg_last_batch(self.h, C.byref(b)))
        return b

    def finalize_begin(self):
        """-> (d_bm_first, d_bm_solid, n_words): this rank's bitmaps over the global sketch, to be summed over ranks"""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self.L.mdbg_finalize_begin(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value or 0, b.value or 0, n.value

    def finalize_end(self):
        """-> (Nodes with device pointers, d_row, n_nodes_global)"""
        nd, row, ng = Nodes(), C.c_void_p(), C.c_uint64()
        self._chk(self.L.mdbg_finalize_end(self.h, C.byref(nd), C.byref(row), C.byref(ng)))
        return nd, row.value or 0, ng.value

    def to_host(self, d_ptr, nbytes, dtype=np.uint8):
        """copy nbytes of device memory into a fresh numpy array"""
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(self.L.mdbg_copy_to_host(self.h, out.ctypes.data, d_ptr, out.nbytes))
        return out

    def sync(self):
        self._chk(self.L.mdbg_sync(self.h))

    def stats(self):
        s = Stats()
        self._chk(self.L.mdbg_get_stats(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_ if f != "reserved"}

    def synth_reads_device(self, seed, genome_len, n_reads, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000, first_read=0):
        """-> (d_bases ptr, d_offsets ptr, n_bases); buffers are owned by the context (valid until the next call)"""
        sp = SynthParams(seed=seed, genome_len=genome_len, n_reads=n_reads, mean_len=mean_len, sd_len=sd_len, min_len=min_len,
                         max_len=max_len, err_ppm=err_ppm)
        db, do, nb = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self.L.mdbg_synth_reads_device(self.h, C.byref(sp), first_read, C.byref(db), C.byref(do), C.byref(nb)))
        return db.value, do.value, nb.value

    def close(self):
        if getattr(self, "h", None):
            self.L.mdbg_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
