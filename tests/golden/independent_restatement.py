#!/usr/bin/env python3
"""A SECOND, independent restatement of rust-mdbg's hot path, written from the reference sources and from the published
definition of ntHash — it imports nothing from oracle/ and shares no code with it.

Why it exists: the reference cannot be built in this image (Rust, no cargo; the `nthash` crate is not vendored) and it has
no tests or golden outputs, so nothing produced BY the reference can pin the C++ oracle ("parity unpinned").
The strongest evidence available is two restatements written separately, in different languages and with different
algorithms, that agree bit for bit:
  * oracle/mdbg_oracle.cpp  — C++, rolling ntHash (the crate's iterator recurrences), per-read loops, hash maps;
  * this file               — Python/numpy, the DIRECT (non-rolling) definition of ntHash evaluated position by position
                              with whole-array operations, run-start detection by array comparison, plain dicts.
`python tests/golden/independent_restatement.py` regenerates tests/golden/independent_cfg1.json (digests of the reference's
example file, BASELINE.json configs[0]) and tests/golden/independent_cases.json (seeded small cases with full expected
outputs); tests/test_oracle_independent.py checks the C++ oracle against both, and the GPU parity tests use the cases too.

Sources followed (paths in the rust-mdbg tree):
  src/read.rs:157-174   encode_rle          src/read.rs:176-211   extract_density (hash_bound :183, filter :196)
  src/kmer_vec.rs:28-39 reverse / normalize  src/main.rs:756-781   window loop, shift, read_offsets
  src/main.rs:632-709   add_kminmer (non-Bloom branch)            src/main.rs:922-929   abundance filter
  src/main.rs:1014-1117 (k-1)-mer index, orientation tests, presimp, overlaps
ntHash (Mohamadi et al. 2016; crate nthash 0.5.x): h(A)=0x3c8bfbb395c60474 h(C)=0x3193c18562a02b4c h(G)=0x20323ed082572324
h(T)=0x295549f54be24456 h(N)=0; forward = XOR_i rol(h(s_i), l-1-i); reverse = XOR_i rol(h(comp s_i), i); canonical = min.
"""
import gzip
import hashlib
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
U64 = np.uint64
SEEDS = {"A": 0x3C8BFBB395C60474, "C": 0x3193C18562A02B4C, "G": 0x20323ED082572324, "T": 0x295549F54BE24456, "N": 0}
COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
HPC_SET = set(b"ACTGactgNn")


def hash_bound(density):
    """src/read.rs:183 — (density as f64 * u64::MAX as f64) as u64: u64::MAX rounds to 2^64 in f64, the cast saturates"""
    v = float(density) * 18446744073709551616.0
    if not v > 0.0:
        return 0
    if v >= 18446744073709551616.0:
        return (1 << 64) - 1
    return int(v)


def run_starts(seq):
    """src/read.rs:157-174 — indices of the bytes encode_rle keeps (the first byte of every run; only bytes of
    "ACTGactgNn" form runs); an empty input keeps the sentinel '#' at position 0, which can never reach length l >= 2"""
    a = np.frombuffer(seq, dtype=np.uint8)
    if len(a) == 0:
        return a, np.zeros(0, dtype=np.int64)
    collapsible = np.isin(a, np.frombuffer(bytes(sorted(HPC_SET)), dtype=np.uint8))
    same_as_prev = np.zeros(len(a), dtype=bool)
    same_as_prev[1:] = (a[1:] == a[:-1]) & collapsible[1:]
    keep = np.nonzero(~same_as_prev)[0]
    return a[keep], keep.astype(np.int64)


def rol_arr(x, r):
    r %= 64
    if r == 0:
        return x
    return (x << U64(r)) | (x >> U64(64 - r))


def canonical_hashes(text, l):
    """direct definition for every l-mer of `text` (uint8 array); None if a byte is outside ACGTN (the crate panics)"""
    lut_f = np.zeros(256, dtype=U64)
    lut_r = np.zeros(256, dtype=U64)
    ok = np.zeros(256, dtype=bool)
    for ch, s in SEEDS.items():
        lut_f[ord(ch)] = s
        lut_r[ord(ch)] = SEEDS[COMPLEMENT[ch]]
        ok[ord(ch)] = True
    n = len(text) - l + 1
    if n <= 0:
        return np.zeros(0, dtype=U64)
    if not ok[text].all():
        return None
    f, r = lut_f[text], lut_r[text]
    fh = np.zeros(n, dtype=U64)
    rh = np.zeros(n, dtype=U64)
    for i in range(l):
        fh ^= rol_arr(f[i:i + n], l - 1 - i)
        rh ^= rol_arr(r[i:i + n], i)
    return np.minimum(fh, rh)


SWITCH_BASE = {"a": "t", "c": "g", "t": "a", "g": "c", "u": "a", "A": "T", "C": "G", "T": "A", "G": "C", "U": "A"}   # src/utils.rs:10-24, anything else -> N


def revcomp_str(s):
    return "".join(SWITCH_BASE.get(c, "N") for c in reversed(s))


def lmer_selection(lines, l, density, cmin=2, cmax=100000):
    """--lmer-counts.  lines: [(lmer str, count)] as in the counts file.  src/main.rs:552-565: the table is keyed by
    min(lmer, revcomp(lmer)), a later line replaces an earlier one.  src/minimizers.rs:53-113 (branch with counts): an l-mer is
    selected iff NOT (count >= max or count <= min) and float(hash) / 2^64 <= density, where hash is the canonical ntHash of its first l
    characters; it and its reverse complement are then both accepted.  -> {lmer str: hash}"""
    table = {}
    for lmer, count in lines:
        rc = revcomp_str(lmer)
        table[lmer if lmer < rc else rc] = count
    out = {}
    for lmer, count in table.items():
        h = canonical_hashes(np.frombuffer(lmer[:l].encode(), dtype=np.uint8), l)
        assert h is not None and len(h) == 1
        hv = int(h[0])
        ratio = 1.0 if (count >= cmax or count <= cmin) else float(hv) / 18446744073709551616.0      # int -> f64 rounds to nearest, like `as f64`
        if ratio <= float(density):
            out[lmer] = hv
            out[revcomp_str(lmer)] = hv
    return out


def sketch_read(seq, l, density, already_hpc=False, lmer_sel=None):
    """src/read.rs:176-211 -> (hashes, raw positions) or None for the nthash panic; lmer_sel: lmer_selection() = --lmer-counts (:200-205)"""
    if already_hpc:
        text = np.frombuffer(seq, dtype=np.uint8)
        pos = np.arange(len(text), dtype=np.int64)
    else:
        text, pos = run_starts(seq)
    if len(text) < l:
        return np.zeros(0, dtype=U64), np.zeros(0, dtype=np.int64)
    h = canonical_hashes(text, l)
    if h is None:
        return None
    sel = np.nonzero(h <= U64(hash_bound(density)))[0]
    if lmer_sel is not None:
        tb = bytes(text)
        sel = np.array([i for i in sel if tb[i:i + l].decode("latin-1") in lmer_sel], dtype=np.int64)
        if len(sel) == 0:
            return np.zeros(0, dtype=U64), np.zeros(0, dtype=np.int64)
    return h[sel], pos[sel]


def wang_hash(key, mask):
    """src/read.rs:43-52 (u64 arithmetic wraps in a release build)"""
    M = (1 << 64) - 1
    key = ((~key & M) + ((key << 21) & M)) & M & mask
    key = key ^ (key >> 24)
    key = (key + ((key << 3) & M) + ((key << 8) & M)) & M & mask
    key = key ^ (key >> 14)
    key = (key + ((key << 2) & M) + ((key << 4) & M)) & M & mask
    key = key ^ (key >> 28)
    key = (key + ((key << 31) & M)) & M & mask
    return key


NT4 = {**{c: i for i, cs in enumerate(("Aa", "Cc", "Gg", "TtUu")) for c in cs.encode()}, 0: 0, 1: 1, 2: 2, 3: 3}


def sketch_read_syncmers(seq, l, s, density, already_hpc=False):
    """src/read.rs:215-352 written as a plain state machine over the (HPC) text: canonical 2-bit l-mer / s-mer values are recomputed
    from the text for every position (no rolling), the tracked minimum follows update_window (read.rs:55-80)"""
    if already_hpc:
        text = np.frombuffer(seq, dtype=np.uint8)
        pos = np.arange(len(text), dtype=np.int64)
    else:
        text, pos = run_starts(seq)
    if len(seq) == 0 and not already_hpc:
        return [], []
    hs, ps = [], []
    if len(text) < l:
        return hs, ps
    v = float(density) * float(4 ** l)
    bound = 0 if not v > 0.0 else ((1 << 64) - 1 if v >= 18446744073709551616.0 else int(v))
    codes = [NT4.get(int(c), 4) for c in text]

    def canon(i, n):                     # canonical value of the n-mer ending at i
        f = r = 0
        for j in range(i - n + 1, i + 1):
            f = (f << 2) | codes[j]
        for j in range(i, i - n, -1):
            r = (r << 2) | (3 - codes[j])
        return min(f, r)
    w = l - s + 1                        # s-mers per l-mer
    t = (w + 1) // 2
    run = 0                              # valid bases since the last reset
    window = []                          # [(hash, position)] of the last <= w s-mers, oldest first
    tracked = None                       # position of the tracked minimum
    for i, c in enumerate(codes):
        if c == 4:
            run, window, tracked = 0, [], None
            continue
        run += 1
        emit = False
        if s == 0:
            emit = run >= l
        elif run >= s:
            h = wang_hash(canon(i, s), (1 << (2 * s)) - 1)
            if len(window) < w - 1:
                window.append((h, i - s + 1))
            elif len(window) == w - 1:
                window.append((h, i - s + 1))
                best = min(x[0] for x in window)
                tracked = next(p for x, p in window if x == best)            # first full window: leftmost minimum
                emit = tracked == window[t - 1][1]
            else:
                gone = window.pop(0)[1]
                window.append((h, i - s + 1))
                cur = dict((p, x) for x, p in window).get(tracked)
                if tracked == gone:
                    best = min(x[0] for x in window)
                    tracked = [p for x, p in window if x == best][-1]       # rescan from the back: rightmost minimum
                elif h < cur:
                    tracked = i - s + 1
                emit = tracked == window[t - 1][1]
        if emit:
            hl = wang_hash(canon(i, l), (1 << (2 * l)) - 1)
            if hl <= bound:
                hs.append(hl)
                ps.append(int(pos[i - l + 1]))
    return hs, ps


class Graph:
    """src/main.rs:632-709,756-781 with --threads 1 and without --bf"""

    def __init__(self, k, l, density, minabund, already_hpc=False, syncmer_s=None, lmer_sel=None):
        self.k, self.l, self.d, self.A, self.hpc_in = k, l, density, minabund, already_hpc
        self.sync_s = syncmer_s
        self.lmer_sel = lmer_sel
        self.nodes = {}          # key tuple -> [index, abundance(u16), seqlen(u32), shift(u16,u16), reversed, src_read, src_start, src_end, shift_full]
        self.next_index = 0
        self.n_minimizers = 0
        self.n_windows = 0
        self.sketches = []

    def add_read(self, ordinal, seq):
        if self.sync_s is not None:
            hs_, ps_ = sketch_read_syncmers(seq, self.l, self.sync_s, self.d, self.hpc_in)
            sk = (np.array(hs_, dtype=U64), np.array(ps_, dtype=np.int64))
        else:
            sk = sketch_read(seq, self.l, self.d, self.hpc_in, self.lmer_sel)
        if sk is None:
            return False
        h, p = sk
        self.sketches.append((h, p))
        self.n_minimizers += len(h)
        k, l, A = self.k, self.l, self.A
        hl = [int(x) for x in h]
        pl = [int(x) for x in p]
        if len(hl) > k:                                                   # strictly more than k (main.rs:756)
            for i in range(len(hl) - k + 1):
                self.n_windows += 1
                win = tuple(hl[i:i + k])
                rev = win[::-1]
                reversed_ = not (win < rev)                               # normalize(): ties go to the reversed copy
                key = rev if reversed_ else win
                a, b = pl[i + 1] - pl[i], pl[i + k - 1] - pl[i + k - 2]
                shift = (b, a) if reversed_ else (a, b)
                offs = (pl[i], pl[i + k - 1] + l, pl[i + k - 1] + 1 - pl[i] + 1)
                meta = [offs[2] & 0xFFFFFFFF, (shift[0] & 0xFFFF, shift[1] & 0xFFFF), reversed_, ordinal, offs[0], offs[1], shift]
                e = self.nodes.get(key)
                if e is None:                                             # first sighting: index, abundance 0 -> 1
                    e = [self.next_index, 0] + meta
                    self.next_index += 1
                    self.nodes[key] = e
                if e[1] == A - 1:                                         # the sighting that lifts it over the filter refreshes the entry
                    e[2:] = meta
                e[1] = (e[1] + 1) & 0xFFFF                                # u16 arithmetic of a release build
        return True

    def finalize(self, presimp=0.01):
        A, k = self.A, self.k
        kept = {key: e for key, e in self.nodes.items() if A <= 1 or e[1] >= A}      # main.rs:922-929
        order = sorted(kept, key=lambda key: kept[key][0])
        rows = [dict(key=list(key), index=kept[key][0], abundance=kept[key][1], seqlen=kept[key][2], shift=list(kept[key][3]),
                     reversed=int(kept[key][4]), src_read=kept[key][5], src_start=kept[key][6], src_end=kept[key][7],
                     shift_full=list(kept[key][8])) for key in order]

        def norm(t):
            r = t[::-1]
            return t if t < r else r
        index = {}
        for key in order:                                                 # main.rs:1017-1038
            for ov in (norm(key[:-1]), norm(key[1:])):
                index.setdefault(ov, []).append(key)
        edges, removed, n_removed = [], set(), 0
        for n1 in order:
            e1 = kept[n1]
            r1 = n1[::-1]
            for ov in (norm(n1[1:]), norm(n1[:-1])):                      # suffix key first (main.rs:1051-1053)
                cands = []
                for n2 in index.get(ov, []):
                    e2 = kept[n2]
                    r2 = n2[::-1]
                    if n1[1:] == n2[:-1]:
                        cands.append((e2, "+", "+"))
                    if n1[1:] == r2[:-1]:
                        cands.append((e2, "+", "-"))
                    if r1[1:] == n2[:-1]:
                        cands.append((e2, "-", "+"))
                    if r1[1:] == r2[:-1]:
                        cands.append((e2, "-", "-"))
                if not cands:
                    continue
                ref = min(max(c[0][1] for c in cands), e1[1])
                for e2, o1, o2 in cands:
                    if presimp > 0.0 and len(cands) >= 2 and np.float32(e2[1]) < np.float32(presimp) * np.float32(ref):
                        n_removed += 1
                        removed.add((e1[0], e2[0]))
                        continue
                    sh = e1[3][0] if o1 == "+" else e1[3][1]
                    ov_len = min((e1[2] - sh) & 0xFFFFFFFF, (e2[2] - 1) & 0xFFFFFFFF)
                    edges.append((e1[0], o1, e2[0], o2, ov_len))
        if presimp > 0.0:
            edges = [e for e in edges if (e[0], e[2]) not in removed and (e[2], e[0]) not in removed]
        return dict(n_minimizers=self.n_minimizers, n_windows=self.n_windows, n_nodes_before=len(self.nodes), n_nodes=len(rows),
                    nodes=rows, edges=sorted(edges), presimp_removed=n_removed)


# ---- digests shared with tests/golden/make_golden.py's format (so the two generators can be compared) -------------
def node_digest(rows):
    ks = sorted(tuple(r["key"]) + (r["abundance"],) for r in rows)
    s = "".join(",".join(map(str, t[:-1])) + ":" + str(t[-1]) + "\n" for t in ks)
    return hashlib.sha256(s.encode()).hexdigest()


def edge_digest(edges):
    s = "".join("L\t%d\t%s\t%d\t%s\t%dM\n" % e for e in sorted(edges))
    return hashlib.sha256(s.encode()).hexdigest()


def read_fasta_gz(path):
    out = []
    with gzip.open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                out.append(line.strip())
    return out


def config1():
    reads = read_fasta_gz(os.path.join(HERE, "reads-0.00.fa.gz"))
    k, l, d, A = 7, 10, 0.0008, 2
    g = Graph(k, l, d, A)
    for i, r in enumerate(reads):
        assert g.add_read(i, r)
    res = g.finalize(0.01)
    hs = np.concatenate([s[0] for s in g.sketches])
    ps = np.concatenate([s[1] for s in g.sketches]).astype(U64)
    off = np.zeros(len(reads) + 1, dtype=U64)
    off[1:] = np.cumsum([len(s[0]) for s in g.sketches])
    return dict(config=dict(k=k, l=l, density=d, minabund=A, file="reads-0.00.fa.gz"), n_reads=len(reads), n_bases=sum(len(r) for r in reads),
                hash_bound=hash_bound(d), n_minimizers=res["n_minimizers"], n_windows=res["n_windows"], n_nodes_before=res["n_nodes_before"],
                n_nodes=res["n_nodes"], n_edges=len(res["edges"]), presimp_removed=res["presimp_removed"],
                read0_first3=[[int(g.sketches[0][1][i]), int(g.sketches[0][0][i])] for i in range(3)],
                minimizers_sha256=hashlib.sha256(hs.tobytes() + ps.tobytes() + off.tobytes()).hexdigest(),
                nodes_sha256=node_digest(res["nodes"]), edges_sha256=edge_digest(res["edges"]))


def random_cases(seed=20260927, n_cases=48):
    rnd = random.Random(seed)
    cases = []
    for ci in range(n_cases):
        genome = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([300, 1200, 4000])))
        if ci % 5 == 1:                                   # homopolymer-rich
            genome = "".join(ch * rnd.choice([1, 1, 2, 3, 9, 40]) for ch in genome[:400])
        if ci % 7 == 3:                                   # palindromic minimizer strings: a sequence followed by its reverse complement
            half = genome[:600]
            genome = half + "".join(COMPLEMENT[c] for c in reversed(half))
        reads = []
        for _ in range(rnd.randint(1, 14)):
            a = rnd.randrange(len(genome))
            b = min(len(genome), a + rnd.choice([0, 5, 40, 300, 900, 2500]))
            r = genome[a:b]
            if rnd.random() < 0.3:
                r = "".join(COMPLEMENT[c] for c in reversed(r))
            if rnd.random() < 0.15 and r:
                p = rnd.randrange(len(r))
                r = r[:p] + "N" * rnd.randint(1, 4) + r[p + 1:]
            reads.append(r)
        if ci % 11 == 4:
            reads.append("ACGTacgtACGT" * 10)           # lower case: error iff the read is long enough
        k = rnd.choice([2, 3, 4, 5, 7, 9])
        l = rnd.choice([2, 3, 5, 8, 10, 12, 14, 16, 21, 31])
        d = rnd.choice([0.002, 0.01, 0.05, 0.2, 0.6, 1.0])
        A = rnd.choice([1, 2, 2, 3])
        hpc_in = rnd.random() < 0.2
        presimp = rnd.choice([0.0, 0.01, 0.5])
        g = Graph(k, l, d, A, hpc_in)
        err_read = None
        for i, r in enumerate(reads):
            if not g.add_read(i, r.encode()):
                err_read = i
                break
        case = dict(reads=reads, k=k, l=l, density=d, minabund=A, already_hpc=hpc_in, presimp=presimp)
        if err_read is not None:
            case["error_read"] = err_read
        else:
            res = g.finalize(presimp)
            case["sketch"] = [[[int(x) for x in s[1]], [int(x) for x in s[0]]] for s in g.sketches]
            case.update({f: res[f] for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed")})
            case["nodes"] = res["nodes"]
            case["edges"] = [list(e) for e in res["edges"]]
        cases.append(case)
    return cases


def syncmer_cases(seed=20260928, n_cases=20):
    """--syncmers -s (src/read.rs:215-352): small s makes ties between s-mer hashes frequent, which is what exercises the tracked minimum"""
    rnd = random.Random(seed)
    cases = []
    for ci in range(n_cases):
        genome = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([400, 1500, 5000])))
        if ci % 4 == 1:                                   # tandem repeats: every window has tied minima
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([2, 3, 5, 7])))
            genome = genome[:200] + unit * 120 + genome[200:600]
        if ci % 5 == 2:
            genome = "".join(ch * rnd.choice([1, 1, 2, 3, 8]) for ch in genome[:600])
        reads = []
        for _ in range(rnd.randint(1, 10)):
            a = rnd.randrange(len(genome))
            r = genome[a:min(len(genome), a + rnd.choice([0, 8, 60, 400, 1500, 4000]))]
            if rnd.random() < 0.3:
                r = "".join(COMPLEMENT[c] for c in reversed(r))
            if rnd.random() < 0.3 and r:
                p = rnd.randrange(len(r))
                r = r[:p] + rnd.choice(["N", "NN", "n", "X", "acgu"]) + r[p + 1:]
            reads.append(r)
        l = rnd.choice([5, 8, 10, 12, 14, 17, 24, 31])
        sm = rnd.choice([0, 1, 2, 3, 4, 5, 8, l])
        sm = min(sm, l, 16)
        if l - sm + 1 > 32:
            sm = l - 31
        k = rnd.choice([2, 3, 4, 6])
        d = rnd.choice([0.01, 0.1, 0.5, 1.0])
        A = rnd.choice([1, 2, 2, 3])
        hpc_in = rnd.random() < 0.25
        presimp = rnd.choice([0.0, 0.01])
        g = Graph(k, l, d, A, hpc_in, syncmer_s=sm)
        for i, r in enumerate(reads):
            assert g.add_read(i, r.encode())
        res = g.finalize(presimp)
        case = dict(reads=reads, k=k, l=l, density=d, minabund=A, already_hpc=hpc_in, presimp=presimp, syncmer_s=sm)
        case["sketch"] = [[[int(x) for x in s_[1]], [int(x) for x in s_[0]]] for s_ in g.sketches]
        case.update({f: res[f] for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed")})
        case["nodes"] = res["nodes"]
        case["edges"] = [list(e) for e in res["edges"]]
        cases.append(case)
    return cases


def lmer_cases(seed=20260929, n_cases=16):
    """--lmer-counts: the counts file is what a k-mer counter would report for the (HPC) reads, perturbed: some l-mers missing, some listed
    in the non-canonical orientation, some twice with different counts, some above / below the thresholds"""
    rnd = random.Random(seed)
    cases = []
    for ci in range(n_cases):
        genome = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([800, 3000, 8000])))
        if ci % 4 == 1:
            genome = "".join(ch * rnd.choice([1, 1, 2, 4]) for ch in genome[:1500])
        reads = []
        for _ in range(rnd.randint(2, 12)):
            a = rnd.randrange(len(genome))
            r = genome[a:min(len(genome), a + rnd.choice([30, 400, 1500, 4000]))]
            if rnd.random() < 0.4:
                r = "".join(COMPLEMENT[c] for c in reversed(r))
            if rnd.random() < 0.25 and r:
                p = rnd.randrange(len(r))
                r = r[:p] + rnd.choice(["N", "NN"]) + r[p + 1:]
            reads.append(r)
        l = rnd.choice([5, 8, 10, 12, 14, 21, 31, 32])
        k = rnd.choice([2, 3, 4, 6])
        d = rnd.choice([0.05, 0.2, 0.5, 1.0])
        A = rnd.choice([1, 2, 2, 3])
        hpc_in = rnd.random() < 0.25
        counts = {}
        for r in reads:
            text = r.encode() if hpc_in else bytes(run_starts(r.encode())[0])
            for i in range(len(text) - l + 1):
                w = text[i:i + l].decode()
                if "N" not in w:
                    counts[w] = counts.get(w, 0) + 1
        lines = []
        for w, c in counts.items():
            u = rnd.random()
            if u < 0.15:
                continue                                  # the counter did not report it
            if u < 0.35:
                w = revcomp_str(w)
            lines.append((w, c + rnd.choice([0, 0, 0, 1, 3])))
            if u > 0.9:
                lines.append((revcomp_str(w), rnd.choice([1, 2, 7])))     # a second line for the same canonical l-mer: the later one wins
        rnd.shuffle(lines)
        cmin, cmax = rnd.choice([(0, 100000), (1, 100000), (2, 100000), (1, 4), (0, 3)])
        sel = lmer_selection(lines, l, d, cmin, cmax)
        g = Graph(k, l, d, A, hpc_in, lmer_sel=sel)
        for i, r in enumerate(reads):
            assert g.add_read(i, r.encode())
        res = g.finalize(0.01)
        case = dict(reads=reads, k=k, l=l, density=d, minabund=A, already_hpc=hpc_in, presimp=0.01, lmer_lines=[[w, c] for w, c in lines], lmer_min=cmin, lmer_max=cmax)
        case["selected"] = sorted([w, h] for w, h in sel.items())
        case["sketch"] = [[[int(x) for x in s_[1]], [int(x) for x in s_[0]]] for s_ in g.sketches]
        case.update({f: res[f] for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "presimp_removed")})
        case["nodes"] = res["nodes"]
        case["edges"] = [list(e) for e in res["edges"]]
        cases.append(case)
    return cases


if __name__ == "__main__":
    c1 = config1()
    json.dump(c1, open(os.path.join(HERE, "independent_cfg1.json"), "w"), indent=1)
    print("config 1:", {f: c1[f] for f in ("n_minimizers", "n_windows", "n_nodes_before", "n_nodes", "n_edges")}, c1["nodes_sha256"][:16])
    cs = random_cases()
    json.dump(dict(generator="tests/golden/independent_restatement.py", seed=20260927, cases=cs), open(os.path.join(HERE, "independent_cases.json"), "w"))
    sc = syncmer_cases()
    json.dump(dict(generator="tests/golden/independent_restatement.py", seed=20260928, cases=sc), open(os.path.join(HERE, "independent_syncmer_cases.json"), "w"))
    lc = lmer_cases()
    json.dump(dict(generator="tests/golden/independent_restatement.py", seed=20260929, cases=lc), open(os.path.join(HERE, "independent_lmer_cases.json"), "w"))
    print(len(lc), "lmer-counts cases,", sum(len(c["selected"]) for c in lc), "selected l-mers,", sum(c["n_minimizers"] for c in lc), "minimizers,", sum(c["n_nodes"] for c in lc), "nodes")
    print(len(sc), "syncmer cases,", sum(c["n_minimizers"] for c in sc), "minimizers,", sum(c["n_nodes"] for c in sc), "nodes")
    print(len(cs), "cases,", sum("error_read" in c for c in cs), "with the alphabet error,", sum(c.get("n_nodes", 0) for c in cs), "nodes,",
          sum(len(c.get("edges", [])) for c in cs), "edges")
    sys.exit(0)
