"""ctypes loader for the CPU oracle (oracle/mdbg_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (rust_mdbg_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "libmdbg_oracle.so")
    src = os.path.join(_HERE, "mdbg_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmdbg_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_hash_bound.restype = C.c_uint64
    L.orc_hash_bound.argtypes = [C.c_double]
    for f in ("orc_ntf64", "orc_ntr64", "orc_ntc64"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u64p]
    L.orc_nthash_iter.restype = C.c_int64
    L.orc_nthash_iter.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u64p]
    L.orc_encode_rle.restype = C.c_uint64
    L.orc_encode_rle.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, u64p]
    L.orc_revcomp.restype = None
    L.orc_revcomp.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
    L.orc_sketch.restype = C.c_void_p
    L.orc_sketch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int]
    L.orc_sketch_syncmers.restype = C.c_void_p
    L.orc_sketch_syncmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_int]
    L.orc_graph_set_syncmers.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_lmer_map_new.restype = C.c_void_p
    L.orc_lmer_map_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, C.c_uint32]
    L.orc_lmer_map_err.restype = C.c_int
    L.orc_lmer_map_err.argtypes = [C.c_void_p]
    L.orc_lmer_map_n.restype = C.c_uint64
    L.orc_lmer_map_n.argtypes = [C.c_void_p]
    L.orc_lmer_map_lmers.restype = u8p
    L.orc_lmer_map_lmers.argtypes = [C.c_void_p]
    L.orc_lmer_map_hashes.restype = u64p
    L.orc_lmer_map_hashes.argtypes = [C.c_void_p]
    L.orc_lmer_map_free.argtypes = [C.c_void_p]
    L.orc_sketch_lmer.restype = C.c_void_p
    L.orc_sketch_lmer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_void_p]
    L.orc_graph_set_lmer_map.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_sketch_err.restype = C.c_int
    L.orc_sketch_err.argtypes = [C.c_void_p]
    L.orc_sketch_n.restype = C.c_uint64
    L.orc_sketch_n.argtypes = [C.c_void_p]
    for f in ("orc_sketch_hashes", "orc_sketch_pos", "orc_sketch_off"):
        getattr(L, f).restype = u64p
        getattr(L, f).argtypes = [C.c_void_p]
    L.orc_sketch_free.argtypes = [C.c_void_p]
    L.orc_graph_new.restype = C.c_void_p
    L.orc_graph_new.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, C.c_int, C.c_float]
    L.orc_graph_ingest.restype = C.c_int
    L.orc_graph_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.orc_graph_finalize.restype = C.c_int
    L.orc_graph_finalize.argtypes = [C.c_void_p, C.c_int]
    L.orc_graph_counter.restype = C.c_uint64
    L.orc_graph_counter.argtypes = [C.c_void_p, C.c_int]
    for f, t in (("keys", u64p), ("index", u32p), ("abundance", u16p), ("seqlen", u32p), ("shift", u16p),
                 ("shift_full", u64p), ("src_read", u64p), ("src_start", u64p), ("src_end", u64p), ("reversed", u8p),
                 ("edge_n1", u32p), ("edge_n2", u32p), ("edge_overlap", u32p), ("edge_o1", u8p), ("edge_o2", u8p),
                 ("seqline_index", u32p), ("seqline_read", u64p), ("seqline_start", u64p), ("seqline_end", u64p),
                 ("seqline_rev", u8p), ("seqline_shift", u64p)):
        fn = getattr(L, "orc_graph_" + f)
        fn.restype = t
        fn.argtypes = [C.c_void_p]
    L.orc_graph_free.argtypes = [C.c_void_p]
    L.orc_graph_from_nodes.restype = C.c_void_p
    L.orc_graph_from_nodes.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    L.orc_count_threaded.restype = C.c_int64
    L.orc_count_threaded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double,
                                     C.c_uint32, C.c_int, C.c_int, u64p]
    L.orc_count_digest_threaded.restype = C.c_int64
    L.orc_count_digest_threaded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double,
                                            C.c_uint32, C.c_int, C.c_int, u64p, u64p]
    _LIB = L
    return L


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def hash_bound(d):
    return int(lib().orc_hash_bound(d))


def ntf64(s, i, k):
    o = C.c_uint64()
    e = lib().orc_ntf64(s, i, k, C.byref(o))
    if e:
        raise ValueError("alphabet")
    return o.value


def ntr64(s, i, k):
    o = C.c_uint64()
    e = lib().orc_ntr64(s, i, k, C.byref(o))
    if e:
        raise ValueError("alphabet")
    return o.value


def ntc64(s, i, k):
    o = C.c_uint64()
    e = lib().orc_ntc64(s, i, k, C.byref(o))
    if e:
        raise ValueError("alphabet")
    return o.value


def nthash_iter(s, k):
    n = len(s)
    out = (C.c_uint64 * max(1, n))()
    r = lib().orc_nthash_iter(s, n, k, out)
    if r < 0:
        raise ValueError("nthash error %d" % r)
    return [int(out[i]) for i in range(r)]


def encode_rle(s):
    n = len(s)
    hpc = C.create_string_buffer(n + 2)
    pos = (C.c_uint64 * (n + 2))()
    m = lib().orc_encode_rle(s, n, hpc, pos)
    return hpc.raw[:m], [int(pos[i]) for i in range(m)]


def revcomp(s):
    out = C.create_string_buffer(len(s) + 1)
    lib().orc_revcomp(s, len(s), out)
    return out.raw[:len(s)]


def concat_reads(reads):
    """list[bytes] -> (uint8 array, uint64 offsets[n+1])"""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offs


class LmerMap:
    """--lmer-counts: the selected l-mers (src/main.rs:544-566 + src/minimizers.rs:53-113) of a counts file given as its lines
    [(lmer bytes, count)]; cmin / cmax: --lmer_counts_min / _max (defaults 2 / 100000, main.rs:447-448)"""

    def __init__(self, lines, l, density, cmin=2, cmax=100000):
        self.L = lib()
        self.l = l
        b, o = concat_reads([x for x, _ in lines])
        cnt = np.ascontiguousarray([c for _, c in lines], dtype=np.uint32)
        self.h = self.L.orc_lmer_map_new(b.ctypes.data, o.ctypes.data, cnt.ctypes.data, len(lines), l, density, cmin, cmax)
        self.err = self.L.orc_lmer_map_err(self.h)

    def selected(self):
        """-> sorted [(lmer bytes, hash)] including both orientations"""
        n = int(self.L.orc_lmer_map_n(self.h))
        raw = _arr(self.L.orc_lmer_map_lmers(self.h), n * self.l, np.uint8).tobytes()
        hs = _arr(self.L.orc_lmer_map_hashes(self.h), n, np.uint64)
        return [(raw[i * self.l:(i + 1) * self.l], int(hs[i])) for i in range(n)]

    def close(self):
        if self.h:
            self.L.orc_lmer_map_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sketch(bases, offsets, l, density, already_hpc=False, syncmer_s=None, lmer_map=None):
    """-> dict(hashes u64[m], pos u64[m], off u64[n+1], err int); syncmer_s: --syncmers -s (src/read.rs:215-352) instead of the density scheme;
    lmer_map: a LmerMap = --lmer-counts"""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    if lmer_map is not None:
        h = L.orc_sketch_lmer(bases.ctypes.data, offsets.ctypes.data, n, l, density, int(already_hpc), lmer_map.h)
    elif syncmer_s is None:
        h = L.orc_sketch(bases.ctypes.data, offsets.ctypes.data, n, l, density, int(already_hpc))
    else:
        h = L.orc_sketch_syncmers(bases.ctypes.data, offsets.ctypes.data, n, l, syncmer_s, density, int(already_hpc))
    try:
        m = L.orc_sketch_n(h)
        return dict(hashes=_arr(L.orc_sketch_hashes(h), m, np.uint64), pos=_arr(L.orc_sketch_pos(h), m, np.uint64),
                    off=_arr(L.orc_sketch_off(h), n + 1, np.uint64), err=L.orc_sketch_err(h))
    finally:
        L.orc_sketch_free(h)


class Graph:
    """Sequential reference semantics (= rust-mdbg --threads 1, no --bf)."""

    def __init__(self, k, l, density, minabund=2, already_hpc=False, presimp=0.01, syncmer_s=None, lmer_map=None):
        self.L = lib()
        self.k = k
        self.h = self.L.orc_graph_new(k, l, density, minabund, int(already_hpc), presimp)
        if lmer_map is not None:
            self.L.orc_graph_set_lmer_map(self.h, lmer_map.h)
        if syncmer_s is not None:
            self.L.orc_graph_set_syncmers(self.h, syncmer_s)
        self.n_ingested = 0

    def ingest(self, bases, offsets, first_read_ordinal=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if first_read_ordinal is None:
            first_read_ordinal = self.n_ingested
        e = self.L.orc_graph_ingest(self.h, bases.ctypes.data, offsets.ctypes.data, n, first_read_ordinal)
        self.n_ingested += n
        return e

    def finalize(self, with_edges=True):
        L, h, k = self.L, self.h, self.k
        e = L.orc_graph_finalize(h, int(with_edges))
        if e:
            raise ValueError("oracle error %d at read %d" % (e, L.orc_graph_counter(h, 8)))
        c = lambda i: int(L.orc_graph_counter(h, i))
        n, ne, ns = c(4), c(5), c(7)
        out = dict(
            n_reads=c(0), n_minimizers=c(1), n_windows=c(2), n_nodes_before=c(3), n_nodes=n, n_edges=ne,
            presimp_removed=c(6),
            keys=_arr(L.orc_graph_keys(h), n * k, np.uint64).reshape(n, k),
            index=_arr(L.orc_graph_index(h), n, np.uint32), abundance=_arr(L.orc_graph_abundance(h), n, np.uint16),
            seqlen=_arr(L.orc_graph_seqlen(h), n, np.uint32), shift=_arr(L.orc_graph_shift(h), 2 * n, np.uint16).reshape(n, 2),
            shift_full=_arr(L.orc_graph_shift_full(h), 2 * n, np.uint64).reshape(n, 2),
            src_read=_arr(L.orc_graph_src_read(h), n, np.uint64), src_start=_arr(L.orc_graph_src_start(h), n, np.uint64),
            src_end=_arr(L.orc_graph_src_end(h), n, np.uint64), reversed=_arr(L.orc_graph_reversed(h), n, np.uint8),
            edge_n1=_arr(L.orc_graph_edge_n1(h), ne, np.uint32), edge_n2=_arr(L.orc_graph_edge_n2(h), ne, np.uint32),
            edge_overlap=_arr(L.orc_graph_edge_overlap(h), ne, np.uint32),
            edge_o1=_arr(L.orc_graph_edge_o1(h), ne, np.uint8), edge_o2=_arr(L.orc_graph_edge_o2(h), ne, np.uint8),
            seqline_index=_arr(L.orc_graph_seqline_index(h), ns, np.uint32), seqline_read=_arr(L.orc_graph_seqline_read(h), ns, np.uint64),
            seqline_start=_arr(L.orc_graph_seqline_start(h), ns, np.uint64), seqline_end=_arr(L.orc_graph_seqline_end(h), ns, np.uint64),
            seqline_rev=_arr(L.orc_graph_seqline_rev(h), ns, np.uint8),
            seqline_shift=_arr(L.orc_graph_seqline_shift(h), 2 * ns, np.uint64).reshape(ns, 2),
        )
        return out

    def close(self):
        if self.h:
            self.L.orc_graph_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def edges_from_nodes(nodes, presimp=0.01):
    """reference emitter (src/main.rs:1014-1117) applied to a node table dict -> sorted list of (n1, o1, n2, o2, overlap)"""
    L = lib()
    k = nodes["keys"].shape[1] if len(nodes["keys"]) else 1
    n = len(nodes["index"])
    keys = np.ascontiguousarray(nodes["keys"], dtype=np.uint64)
    idx = np.ascontiguousarray(nodes["index"], dtype=np.uint32)
    ab = np.ascontiguousarray(nodes["abundance"], dtype=np.uint16)
    sl = np.ascontiguousarray(nodes["seqlen"], dtype=np.uint32)
    sh = np.ascontiguousarray(nodes["shift"], dtype=np.uint16)
    h = L.orc_graph_from_nodes(k, n, keys.ctypes.data, idx.ctypes.data, ab.ctypes.data, sl.ctypes.data, sh.ctypes.data, presimp)
    try:
        assert L.orc_graph_finalize(h, 1) == 0
        ne = int(L.orc_graph_counter(h, 5))
        e = zip(_arr(L.orc_graph_edge_n1(h), ne, np.uint32).tolist(), _arr(L.orc_graph_edge_o1(h), ne, np.uint8).tolist(),
                _arr(L.orc_graph_edge_n2(h), ne, np.uint32).tolist(), _arr(L.orc_graph_edge_o2(h), ne, np.uint8).tolist(),
                _arr(L.orc_graph_edge_overlap(h), ne, np.uint32).tolist())
        return sorted(e), int(L.orc_graph_counter(h, 6))
    finally:
        L.orc_graph_free(h)


def count_threaded(bases, offsets, k, l, density, minabund=2, already_hpc=False, threads=1):
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    w = C.c_uint64()
    r = L.orc_count_threaded(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, l, density, minabund,
                             int(already_hpc), threads, C.byref(w))
    if r < 0:
        raise ValueError("oracle error %d" % r)
    return int(r), int(w.value)


def count_digest_threaded(bases, offsets, k, l, density, minabund=2, already_hpc=False, threads=1):
    """-> (solid nodes, window occurrences, (sum, xor)): the counts of count_threaded plus the order-free digest of the filtered node table {(key, abundance)}
    (mdbg_oracle.cpp, orc_node_hash)"""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    w = C.c_uint64()
    dg = (C.c_uint64 * 2)()
    r = L.orc_count_digest_threaded(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, l, density, minabund,
                                    int(already_hpc), threads, C.byref(w), dg)
    if r < 0:
        raise ValueError("oracle error %d" % r)
    return int(r), int(w.value), (int(dg[0]), int(dg[1]))


def nodes_digest(keys, abundance):
    """the same digest from a node table held as arrays (keys: n x k u64, abundance: n u16) — plain numpy, for tests"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = len(abundance)
    keys = keys.reshape(n, -1) if n else keys.reshape(0, 1)
    M = (1 << 64) - 1

    def fmix(x):
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
        return x ^ (x >> np.uint64(33))
    with np.errstate(over="ignore"):
        h = np.uint64(0x243F6A8885A308D3) ^ np.asarray(abundance, dtype=np.uint16).astype(np.uint64)
        for j in range(keys.shape[1]):
            h = fmix(h ^ keys[:, j])
        return (int(h.sum(dtype=np.uint64)) & M if n else 0, int(np.bitwise_xor.reduce(h)) if n else 0)
