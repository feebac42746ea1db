"""CPU stand-in for GpuEngine (tests only): same stage interface as rust_mdbg_amd.dist.GpuEngine, with the sketch taken
from the oracle and everything else in plain Python/numpy.  Lets the multi-rank driver logic (routing, ordering,
cross-rank resolution) run under gloo / threads without a GPU."""
import hashlib

import numpy as np
import torch

from oracle import oracle as O

WIN_BITS = 26


def _i64(x):
    return np.asarray(x, dtype=np.uint64).view(np.int64)


class NumpyEngine:
    def __init__(self, k, l, density, minabund):
        self.k, self.l, self.d, self.A = k, l, density, minabund
        self.device = torch.device("cpu")
        self.reset()

    def reset(self):
        self.ranges, self.batches, self.pending = [], [], []
        self.table = {}          # key tuple -> [count, sorted ordinals]
        self.keys_list = []

    def sketch_host(self, bases, offsets, first_ordinal):
        sk = O.sketch(bases, offsets, self.l, self.d)
        assert sk["err"] == 0
        b = dict(first=int(first_ordinal), n=len(offsets) - 1, sk=sk)
        self.batches.append(b)
        self.pending.append(b)
        self.ranges.append((int(first_ordinal), len(offsets) - 1))

    def _windows(self, b):
        sk, k = b["sk"], self.k
        for r in range(b["n"]):
            lo, hi = int(sk["off"][r]), int(sk["off"][r + 1])
            if hi - lo > k:
                for i in range(hi - lo - k + 1):
                    w = [int(x) for x in sk["hashes"][lo + i:lo + i + k]]
                    rev = not (w < w[::-1])
                    yield tuple(w[::-1] if rev else w), ((b["first"] + r) << WIN_BITS) | i

    def route_pack(self, world):
        buckets = [[] for _ in range(world)]
        for b in self.pending:
            for key, ordn in self._windows(b):
                h = int.from_bytes(hashlib.blake2b(np.asarray(key, dtype=np.uint64).tobytes(), digest_size=8).digest(), "little")
                buckets[(h * world) >> 64].append(list(key) + [ordn, h])
        self.pending = []
        counts = [len(x) for x in buckets]
        rows = [r for bk in buckets for r in bk]
        arr = _i64(np.asarray(rows, dtype=np.uint64).reshape(len(rows), self.k + 2))
        return torch.from_numpy(arr.copy()), counts

    def insert_records(self, recs):
        a = recs.numpy().view(np.uint64)
        for row in a:
            key = tuple(int(x) for x in row[:self.k])
            e = self.table.get(key)
            if e is None:
                e = self.table[key] = [0, []]
                self.keys_list.append(key)
            e[0] += 1
            e[1].append(int(row[self.k]))

    def export(self, world, span_lo, span_rank):
        """the two query lists of GpuEngine.export, bucketed by the rank whose read span holds the ordinal"""
        import bisect

        def owner(ordn):
            return span_rank[bisect.bisect_right(span_lo, ordn >> WIN_BITS) - 1]
        ents = []
        for slot, key in enumerate(self.keys_list):
            c, ords = self.table[key]
            ords = sorted(ords)
            solid = self.A == 1 or (c & 0xFFFF) >= self.A
            ents.append((ords[0], solid, ords[self.A - 1] if solid else None, c, slot))
        order_a = sorted(range(len(ents)), key=lambda i: owner(ents[i][0]))
        pos_a = {i: p for p, i in enumerate(order_a)}
        sol = [i for i in range(len(ents)) if ents[i][1]]
        order_s = sorted(sol, key=lambda i: owner(ents[i][2]))
        cnt_a, cnt_s = [0] * world, [0] * world
        for i in order_a:
            cnt_a[owner(ents[i][0])] += 1
        for i in order_s:
            cnt_s[owner(ents[i][2])] += 1
        T = lambda v: torch.from_numpy(_i64(v).copy()) if len(v) else torch.empty(0, dtype=torch.int64)
        return dict(first=T([ents[i][0] for i in order_a]), solid=torch.tensor([int(ents[i][1]) for i in order_a], dtype=torch.int64),
                    counts_all=cnt_a, ath=T([ents[i][2] for i in order_s]), count=torch.tensor([ents[i][3] for i in order_s], dtype=torch.int64),
                    slot=torch.tensor([ents[i][4] for i in order_s], dtype=torch.int64),
                    idx_all=torch.tensor([pos_a[i] for i in order_s], dtype=torch.int64), counts_solid=cnt_s)

    # ---- replicated-sketch mode -------------------------------------------------------------------------
    def set_partition(self, world, rank):
        self.world, self.rank = world, rank

    def ingest_sketch(self, hashes, pos, read_off, first_ordinal):
        sk = dict(hashes=hashes.numpy().view(np.uint64).copy(), pos=pos.numpy().astype(np.uint64), off=read_off.numpy().view(np.uint64).copy())
        b = dict(first=int(first_ordinal), n=len(sk["off"]) - 1, sk=sk)
        self.batches.append(b)
        self.pending.append(b)

    def last_sketch(self):
        b = self.batches[-1]
        sk = b["sk"]
        return (torch.from_numpy(_i64(sk["hashes"]).copy()), torch.from_numpy(np.asarray(sk["pos"]).astype(np.int32)),
                torch.from_numpy(_i64(sk["off"]).copy()), b["first"], b["n"])

    def reserve_import(self, sizes):
        out = []
        for m in sizes:
            h, p = torch.empty(m, dtype=torch.int64), torch.empty(m, dtype=torch.int32)
            out.append((h, p, (h, p)))           # the token of this engine is simply the pair of receive buffers
        return out

    def commit_import(self, token, rel_off, first_ordinal, owned=None, window_list=None):
        self.ingest_sketch(token[0], token[1], rel_off, first_ordinal)
        if window_list is not None:            # the sender listed this rank's windows: pairs (index of the first minimizer, read), relative to the batch
            self.batches[-1]["list"] = window_list.numpy().astype(np.int64).reshape(-1, 2)
            self.batches[-1]["owned"] = owned

    def _owner(self, w):
        k, M = self.k, (1 << 64) - 1
        x = (w[0] + w[k - 1] + w[(k - 1) >> 1] + w[k >> 1]) & M
        return (self._fmix(x) * self.world) >> 64

    def owner_lists(self, world):
        """the mdbg_owner_lists of this engine: windows of the batch sketched last per owning rank -> (counts, int32 tensor of (window, read)
        pairs bucketed by owner)"""
        assert world == self.world
        b = self.batches[-1]
        sk, k = b["sk"], self.k
        buckets = [[] for _ in range(world)]
        for r in range(b["n"]):
            lo, hi = int(sk["off"][r]), int(sk["off"][r + 1])
            if hi - lo > k:
                for i in range(hi - lo - k + 1):
                    w = [int(x) for x in sk["hashes"][lo + i:lo + i + k]]
                    buckets[self._owner(w)].append((lo + i, r))
        flat = np.array([v for bk in buckets for pr in bk for v in pr], dtype=np.int32)
        return [len(bk) for bk in buckets], torch.from_numpy(flat)

    @staticmethod
    def _fmix(x):
        M = (1 << 64) - 1
        x ^= x >> 33; x = x * 0xff51afd7ed558ccd & M; x ^= x >> 33; x = x * 0xc4ceb9fe1a85ec53 & M; x ^= x >> 33
        return x

    def insert_owned(self):
        k, M = self.k, (1 << 64) - 1
        for b in self.pending:
            sk = b["sk"]
            if "list" in b:                     # exactly the listed windows (each one checked, like insert_listed_span_kernel), and their number
                cand, n_ok = [(int(r), int(wi) - int(sk["off"][int(r)])) for wi, r in b["list"]], 0
                for r, i in cand:
                    lo, hi = int(sk["off"][r]), int(sk["off"][r + 1])
                    assert 0 <= i and hi - lo > k and lo + i + k <= hi
                    n_ok += 1
                assert b["owned"] is None or n_ok == b["owned"]
            else:
                cand = [(r, i) for r in range(b["n"]) if int(sk["off"][r + 1]) - int(sk["off"][r]) > k
                        for i in range(int(sk["off"][r + 1]) - int(sk["off"][r]) - k + 1)]
            for r, i in cand:
                lo = int(sk["off"][r])
                if True:
                    if True:
                        w = [int(x) for x in sk["hashes"][lo + i:lo + i + k]]
                        x = (w[0] + w[k - 1] + w[(k - 1) >> 1] + w[k >> 1]) & M
                        if (self._fmix(x) * self.world) >> 64 != self.rank:
                            assert "list" not in b, "a listed window is not owned by this rank"
                            continue
                        rev = not (w < w[::-1])
                        key = tuple(w[::-1] if rev else w)
                        ent = self.table.get(key)
                        if ent is None:
                            ent = self.table[key] = [0, []]
                            self.keys_list.append(key)
                        ent[0] += 1
                        ent[1].append(((b["first"] + r) << WIN_BITS) | i)
        self.pending = []

    def _dense(self, ordn):
        ro, win = ordn >> WIN_BITS, ordn & ((1 << WIN_BITS) - 1)
        base = 0
        for b in sorted(self.batches, key=lambda q: q["first"]):
            if b["first"] <= ro < b["first"] + b["n"]:
                return base + int(b["sk"]["off"][ro - b["first"]]) + win
            base += len(b["sk"]["hashes"])
        raise KeyError(ordn)

    def finalize_begin(self):
        total = sum(len(b["sk"]["hashes"]) for b in self.batches)
        bf = np.zeros((total + 63) // 64, dtype=np.uint64)
        bs = np.zeros_like(bf)
        self._fin = []
        for key in self.keys_list:
            c, ords = self.table[key]
            ords = sorted(ords)
            solid = self.A == 1 or (c & 0xFFFF) >= self.A
            D = self._dense(ords[0])
            bf[D >> 6] |= np.uint64(1 << (D & 63))
            if solid:
                bs[D >> 6] |= np.uint64(1 << (D & 63))
                self._fin.append((key, c, D, ords[self.A - 1]))
        self._bf, self._bs = torch.from_numpy(bf.view(np.int64)), torch.from_numpy(bs.view(np.int64))
        return self._bf, self._bs

    def finalize_end(self):
        bf, bs = self._bf.numpy().view(np.uint64), self._bs.numpy().view(np.uint64)
        bits_f = np.unpackbits(bf.view(np.uint8), bitorder="little")
        bits_s = np.unpackbits(bs.view(np.uint8), bitorder="little")
        cf, cs = np.concatenate([[0], np.cumsum(bits_f)]), np.concatenate([[0], np.cumsum(bits_s)])
        keys, rows, idx, ab, metas = [], [], [], [], []
        for key, c, D, oa in self._fin:
            keys.append(key); rows.append(int(cs[D])); idx.append(int(cf[D])); ab.append(c & 0xFFFF); metas.append(oa)
        meta = self.resolve_meta(torch.from_numpy(_i64(metas).copy()) if metas else torch.empty(0, dtype=torch.int64))
        n = len(keys)
        T = lambda v: torch.tensor(v, dtype=torch.int64)
        return dict(keys=torch.from_numpy(_i64(np.asarray(keys, dtype=np.uint64).reshape(n, self.k)).copy()), index=T(idx), row=T(rows), abundance=T(ab),
                    seqlen=meta[:, 0] & 0xFFFFFFFF, reversed=(meta[:, 0] >> 32) & 1, shift_full=meta[:, 1:3], src_read=meta[:, 3], src_start=meta[:, 4],
                    src_end=meta[:, 5], n_nodes=int(bits_s.sum()), n_nodes_before=int(bits_f.sum()), n_local=n)

    def resolve_first(self, ords, solid):
        o = ords.numpy().view(np.uint64)
        s = solid.numpy().astype(bool)
        order = np.argsort(o, kind="stable")
        rf = np.empty(len(o), dtype=np.int64)
        rf[order] = np.arange(len(o))
        rs = np.empty(len(o), dtype=np.int64)
        rs[order] = np.cumsum(s[order]) - s[order]
        return torch.from_numpy(rf), torch.from_numpy(rs), int(len(o)), int(s.sum())

    def _locate(self, ordn):
        ro, win = ordn >> WIN_BITS, ordn & ((1 << WIN_BITS) - 1)
        for b in self.batches:
            if b["first"] <= ro < b["first"] + b["n"]:
                return b["sk"], int(b["sk"]["off"][ro - b["first"]]) + win
        raise KeyError(ordn)

    def resolve_meta(self, ords):
        out = np.zeros((len(ords), 6), dtype=np.uint64)
        k, l = self.k, self.l
        for q, ordn in enumerate(ords.numpy().view(np.uint64).tolist()):
            if ordn == (1 << 64) - 1:
                continue
            sk, i = self._locate(ordn)
            w = [int(x) for x in sk["hashes"][i:i + k]]
            p = [int(x) for x in sk["pos"][i:i + k]]
            rev = not (w < w[::-1])
            first, last = p[1] - p[0], p[k - 1] - p[k - 2]
            out[q] = [((p[k - 1] + 1 - p[0] + 1) & 0xFFFFFFFF) | (int(rev) << 32), last if rev else first, first if rev else last,
                      ordn >> WIN_BITS, p[0], p[k - 1] + l]
        return torch.from_numpy(out.view(np.int64).copy())

    def keys(self, slots):
        arr = np.asarray([self.keys_list[int(s)] for s in slots.tolist()], dtype=np.uint64).reshape(len(slots), self.k)
        return torch.from_numpy(arr.view(np.int64).copy())
