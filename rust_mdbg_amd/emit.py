"""ctypes binding of include/mdbg_emit.h — the host-side emitter rust-mdbg keeps after the hot path
(edges + presimp, GFA S/L lines, .sequences in an LZ4 frame; src/main.rs:1006-1121, 614-630, 693-708)."""
import ctypes as C
import os

import numpy as np

from .api import Nodes

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


from .api import EdgeList as Edges          # one type for the host emitter's and the GPU's edge list (include/mdbg_hip.h)


EXPORTS = ["mdbg_lmer_filter_from_counts", "mdbg_lmer_filter_free", "mdbg_packed_words", "mdbg_pack_reads", "mdbg_seqfile_write_batch_part", "mdbg_emit_create", "mdbg_emit_destroy", "mdbg_emit_edges", "mdbg_emit_write_gfa", "mdbg_seqfile_open",
           "mdbg_seqfile_write_batch", "mdbg_seqfile_close"]


def load_library():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "libmdbg_emit.so")
        if not os.path.exists(p):
            raise ImportError("libmdbg_emit.so not built; run `make -C rust_mdbg_amd/csrc`")
        L = C.CDLL(p)
        vp = C.c_void_p
        L.mdbg_emit_create.restype = vp
        L.mdbg_emit_destroy.argtypes = [vp]
        L.mdbg_emit_edges.argtypes = [vp, C.POINTER(Nodes), C.c_float, C.POINTER(Edges)]
        L.mdbg_emit_write_gfa.argtypes = [C.c_char_p, C.POINTER(Nodes), C.POINTER(Edges)]
        L.mdbg_seqfile_open.restype = vp
        L.mdbg_seqfile_open.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
        L.mdbg_seqfile_write_batch.argtypes = [vp, C.POINTER(Nodes), vp, vp, C.c_uint64, C.c_uint64]
        L.mdbg_seqfile_close.argtypes = [vp]
        L.mdbg_seqfile_write_batch_part.argtypes = [vp, C.POINTER(Nodes), C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, C.c_uint64]
        L.mdbg_packed_words.restype = C.c_uint64
        L.mdbg_packed_words.argtypes = [C.c_uint64]
        L.mdbg_pack_reads.argtypes = [vp, C.c_uint64, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
        L.mdbg_lmer_filter_from_counts.argtypes = [C.c_char_p, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint64)),
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mdbg_lmer_filter_free.argtypes = [vp]
        L.mdbg_lmer_filter_free.restype = None
        _LIB = L
    return _LIB


class NodeTable:
    """host node table (dict of numpy arrays as returned by Mdbg.finalize) viewed as a C `mdbg_nodes`"""

    def __init__(self, nodes):
        # (a table from Mdbg.finalize(gfa_only=True) holds index / seqlen / abundance only: the other pointers are null, write_gfa reads none of them)
        self.keep = {f: (None if nodes.get(f) is None else np.ascontiguousarray(nodes[f], dtype=t)) for f, t in (
            ("keys", np.uint64), ("index", np.uint32), ("abundance", np.uint16), ("seqlen", np.uint32), ("shift", np.uint16),
            ("shift_full", np.uint64), ("src_read", np.uint64), ("src_start", np.uint64), ("src_end", np.uint64), ("reversed", np.uint8))}
        self.gfa_only = self.keep["keys"] is None
        k = self.keep["keys"].shape[1] if (not self.gfa_only and self.keep["keys"].ndim == 2) else int(nodes.get("k", 0))
        n = len(self.keep["index"])
        P = lambda f, t: self.keep[f].ctypes.data_as(C.POINTER(t)) if self.keep[f] is not None else None
        self.c = Nodes(n=n, k=k, keys=P("keys", C.c_uint64), index=P("index", C.c_uint32), abundance=P("abundance", C.c_uint16),
                       seqlen=P("seqlen", C.c_uint32), shift=P("shift", C.c_uint16), shift_full=P("shift_full", C.c_uint64),
                       src_read=P("src_read", C.c_uint64), src_start=P("src_start", C.c_uint64), src_end=P("src_end", C.c_uint64),
                       reversed=P("reversed", C.c_uint8), n_distinct=int(nodes.get("n_nodes_before", 0)), n_wrapped=int(nodes.get("n_wrapped", 0)))


class Emitter:
    def __init__(self):
        self.L = load_library()
        self.h = self.L.mdbg_emit_create()
        self._edges = None

    def edges(self, nodes, presimp=0.01):
        """-> dict(n1, o1, n2, o2, overlap, presimp_removed) — the L-lines of the graph"""
        self.nt = nodes if isinstance(nodes, NodeTable) else NodeTable(nodes)
        if self.nt.gfa_only:
            raise ValueError("a gfa_only node table holds no minimizer lists: edges come from Mdbg.graph_edges")
        e = Edges()
        rc = self.L.mdbg_emit_edges(self.h, C.byref(self.nt.c), presimp, C.byref(e))
        if rc:
            raise RuntimeError("mdbg_emit_edges failed: %d" % rc)
        self._edges = e
        g = lambda p, t: np.ctypeslib.as_array(p, shape=(e.n,)).astype(t, copy=True) if e.n else np.zeros(0, t)
        return dict(n1=g(e.n1, np.uint32), o1=g(e.o1, np.uint8), n2=g(e.n2, np.uint32), o2=g(e.o2, np.uint8), overlap=g(e.overlap, np.uint32),
                    presimp_removed=int(e.presimp_removed))

    def write_gfa(self, path, nodes=None, edges=None):
        """edges: an EdgeList with HOST arrays (Mdbg.graph_edges(raw=True)); default: the list of the last edges() call"""
        nt = self.nt if nodes is None else (nodes if isinstance(nodes, NodeTable) else NodeTable(nodes))
        ed = edges if edges is not None else self._edges
        rc = self.L.mdbg_emit_write_gfa(path.encode(), C.byref(nt.c), C.byref(ed) if ed is not None else None)
        if rc:
            raise RuntimeError("mdbg_emit_write_gfa failed: %d" % rc)

    def write_sequences(self, path, nodes, l, batches):
        """batches: iterable of (bases u8 array, offsets u64 array, first_read_ordinal) — what was ingested"""
        nt = nodes if isinstance(nodes, NodeTable) else NodeTable(nodes)
        if nt.gfa_only:
            raise ValueError("a gfa_only node table cannot be written as .sequences")
        err = C.c_int()
        f = self.L.mdbg_seqfile_open(path.encode(), nt.c.k, l, C.byref(err))
        if not f:
            raise RuntimeError("mdbg_seqfile_open failed: %d" % err.value)
        try:
            for bases, offsets, first in batches:
                bases = np.ascontiguousarray(bases, dtype=np.uint8)
                offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
                rc = self.L.mdbg_seqfile_write_batch(f, C.byref(nt.c), bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, first)
                if rc:
                    raise RuntimeError("mdbg_seqfile_write_batch failed: %d" % rc)
        finally:
            rc = self.L.mdbg_seqfile_close(f)
        if rc:
            raise RuntimeError("mdbg_seqfile_close failed: %d" % rc)

    def write_sequences_parallel(self, prefix, nodes, l, batches, threads):
        """`threads` files "<prefix>.<t>.sequences" (the reference's one-file-per-worker layout, src/main.rs:614-630) written by as many
        threads: every batch is handed to all of them, thread t writes the lines of the nodes i with i % threads == t.  -> the paths"""
        from concurrent.futures import ThreadPoolExecutor
        nt = nodes if isinstance(nodes, NodeTable) else NodeTable(nodes)
        if nt.gfa_only:
            raise ValueError("a gfa_only node table cannot be written as .sequences")
        paths = ["%s.%d.sequences" % (prefix, t) for t in range(threads)]
        files, err = [], C.c_int()
        try:
            for p in paths:
                f = self.L.mdbg_seqfile_open(p.encode(), nt.c.k, l, C.byref(err))
                if not f:
                    raise RuntimeError("mdbg_seqfile_open failed: %d" % err.value)
                files.append(f)
            with ThreadPoolExecutor(max_workers=threads) as pool:
                for bases, offsets, first in batches:
                    bases = np.ascontiguousarray(bases, dtype=np.uint8)
                    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
                    job = lambda t: self.L.mdbg_seqfile_write_batch_part(files[t], C.byref(nt.c), t, threads, bases.ctypes.data, offsets.ctypes.data,
                                                                         len(offsets) - 1, first)      # ctypes releases the GIL
                    for rc in pool.map(job, range(threads)):
                        if rc:
                            raise RuntimeError("mdbg_seqfile_write_batch_part failed: %d" % rc)
        finally:
            rcs = [self.L.mdbg_seqfile_close(f) for f in files]
        if any(rcs):
            raise RuntimeError("mdbg_seqfile_close failed: %s" % rcs)
        return paths

    def close(self):
        if self.h:
            self.L.mdbg_emit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host ingest (include/mdbg_emit.h: mdbg_reader_*) -----------------------------------------------------
READER_EXPORTS = ["mdbg_reader_open", "mdbg_reader_open_mt", "mdbg_reader_next", "mdbg_reader_next_packed", "mdbg_reader_is_fasta", "mdbg_reader_is_parallel",
                  "mdbg_reader_set_allocator", "mdbg_reader_close"]


class Reader:
    """FASTA/FASTQ(.gz) -> batches in the layout of Mdbg.ingest; format by file name like src/main.rs:461-467"""

    def __init__(self, path, strip_newlines=False, threads=1, device_buffers=False):
        """threads > 1: uncompressed files are mapped and parsed by that many threads (mdbg_reader_open_mt).
        device_buffers: the batch buffers come from mdbg_host_alloc of libmdbg_hip.so (page-locked by the first ingest call that sees them: the copy to the
        device is one DMA); needs the GPU library, so only a caller that ingests asks for it"""
        L = load_library()
        L.mdbg_reader_open.restype = C.c_void_p
        L.mdbg_reader_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.mdbg_reader_open_mt.restype = C.c_void_p
        L.mdbg_reader_open_mt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.mdbg_reader_next.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.mdbg_reader_is_fasta.argtypes = [C.c_void_p]
        L.mdbg_reader_close.argtypes = [C.c_void_p]
        L.mdbg_reader_close.restype = None
        self.L = L
        err = C.c_int()
        self.h = L.mdbg_reader_open_mt(path.encode(), int(strip_newlines), int(threads), C.byref(err))
        if not self.h:
            raise OSError("cannot open %s (err %d)" % (path, err.value))
        self.is_fasta = bool(L.mdbg_reader_is_fasta(self.h))
        if device_buffers:
            from .api import load_library as load_hip
            H = load_hip()
            L.mdbg_reader_set_allocator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            rc = L.mdbg_reader_set_allocator(self.h, C.cast(H.mdbg_host_alloc, C.c_void_p), C.cast(H.mdbg_host_free, C.c_void_p))
            if rc:
                raise RuntimeError("mdbg_reader_set_allocator failed: %d" % rc)
        L.mdbg_reader_is_parallel.argtypes = [C.c_void_p]
        self.parallel = bool(L.mdbg_reader_is_parallel(self.h))      # True: batches(copy=False) views stay valid while the NEXT batch is read

    def batches(self, max_bases=256 << 20, copy=True):
        """yields (bases uint8[:], offsets uint64[n+1]), whole records, <= max_bases each; copy=False: views into the reader's buffers,
        valid until the next batch is asked for"""
        while True:
            b, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
            rc = self.L.mdbg_reader_next(self.h, max_bases, C.byref(b), C.byref(o), C.byref(n))
            if rc:
                raise RuntimeError("mdbg_reader_next failed: %d" % rc)
            if n.value == 0:
                return
            offs = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint64)), shape=(n.value + 1,))
            nb = int(offs[-1])
            bases = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb] if b.value else np.zeros(0, np.uint8)
            yield (bases.copy(), offs.copy()) if copy else (bases, offs)

    def batches_packed(self, max_bases=256 << 20, copy=True):
        """yields dicts in the layout of pack_reads / Mdbg.ingest_packed (words, offsets, exc_pos, exc_val, n_bases): the reader packs while it
        parses (mdbg_reader_next_packed).  copy=False: views into the reader's buffers, valid until the batch after the next one is asked for"""
        from .api import PackedBatch
        self.L.mdbg_reader_next_packed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(PackedBatch)]
        while True:
            pb = PackedBatch()
            rc = self.L.mdbg_reader_next_packed(self.h, max_bases, C.byref(pb))
            if rc:
                raise RuntimeError("mdbg_reader_next_packed failed: %d" % rc)
            n = int(pb.n_reads)
            if n == 0:
                return
            offs = np.ctypeslib.as_array(C.cast(pb.offsets, C.POINTER(C.c_uint64)), shape=(n + 1,))
            nb = int(offs[-1])
            nw, ne = (nb + 31) // 32, int(pb.n_exc)
            words = np.ctypeslib.as_array(C.cast(pb.words, C.POINTER(C.c_uint64)), shape=(max(nw, 1),))[:nw]
            ep = np.ctypeslib.as_array(C.cast(pb.exc_pos, C.POINTER(C.c_uint64)), shape=(max(ne, 1),))[:ne] if ne else np.zeros(0, np.uint64)
            ev = np.ctypeslib.as_array(C.cast(pb.exc_val, C.POINTER(C.c_uint8)), shape=(max(ne, 1),))[:ne] if ne else np.zeros(0, np.uint8)
            d = dict(words=words, offsets=offs, exc_pos=ep, exc_val=ev, n_bases=nb)
            yield {f: (v.copy() if copy and hasattr(v, "copy") else v) for f, v in d.items()}

    def close(self):
        if self.h:
            self.L.mdbg_reader_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def pack_reads(bases, offsets, threads=1, exc_cap=None, words_buf=None):
    """ASCII batch -> mdbg_packed_batch arrays (host): dict(words, offsets, exc_pos, exc_val, n_bases).  words_buf: a uint64 array to pack
    into when it is large enough (a caller that packs batch after batch reuses it: fresh pages cost more than the packing itself)"""
    L = load_library()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(bases)
    nw = int(L.mdbg_packed_words(n))
    words = words_buf[:nw] if words_buf is not None and len(words_buf) >= nw else np.empty(nw, dtype=np.uint64)      # the packer writes every word
    cap = 1024 if exc_cap is None else exc_cap
    while True:
        ep, ev, ne = np.zeros(cap, np.uint64), np.zeros(cap, np.uint8), C.c_uint64()
        rc = L.mdbg_pack_reads(bases.ctypes.data, n, words.ctypes.data, ep.ctypes.data, ev.ctypes.data, cap, C.byref(ne), threads)
        if rc == 0:
            break
        if rc != -3 or exc_cap is not None:
            raise RuntimeError("mdbg_pack_reads failed: %d" % rc)
        cap = int(ne.value)
    k = int(ne.value)
    return dict(words=words, offsets=offsets, exc_pos=np.ascontiguousarray(ep[:k]), exc_val=np.ascontiguousarray(ev[:k]), n_bases=n)


def lmer_filter_from_counts(path, l, density, count_min=2, count_max=100000):
    """--lmer-counts FILE (src/main.rs:544-575, src/minimizers.rs:53-113) -> (uint64 codes of the selected l-mers, both orientations, for
    Mdbg.set_lmer_filter; number of lines that cannot match any read l-mer)"""
    L = load_library()
    p, n, ign = C.POINTER(C.c_uint64)(), C.c_uint64(), C.c_uint64()
    rc = L.mdbg_lmer_filter_from_counts(os.fsencode(path), l, density, count_min, count_max, C.byref(p), C.byref(n), C.byref(ign))
    if rc:
        raise RuntimeError("mdbg_lmer_filter_from_counts failed: %d" % rc)
    try:
        codes = np.ctypeslib.as_array(p, shape=(int(n.value),)).astype(np.uint64, copy=True) if n.value else np.zeros(0, dtype=np.uint64)
    finally:
        L.mdbg_lmer_filter_free(p)
    return codes, int(ign.value)
