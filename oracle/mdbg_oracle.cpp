// mdbg_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A plain, single-threaded C++17 restatement of rust-mdbg's per-read hot path, written to be
// obviously equal to the reference rather than fast.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product path (rust_mdbg_amd/) never does.
//
// PARITY STATUS: **parity unpinned** by the reference itself — the reference (Rust) has no tests,
// no golden outputs, cannot be compiled here (no cargo/rustc), and its ntHash arithmetic lives in
// the un-vendored third-party crate `nthash` (Cargo.toml:26 `nthash = "*"`, no lockfile; newest
// release at reference time: 0.5.1).  The oracle is anchored instead on
//   (1) the nthash crate's published known-answer vectors (tests/test_oracle_kat.py), and
//   (2) the reference's own call sites, restated line by line below (file:line cited per function), and
//   (3) for ONE function something the reference itself produced: encode_rle against the outputs of the reference's Python helper
//       utils/remove_homopoly.py (the same rule and the same literal "ACTGactgNn" as src/read.rs:157-174), run in the build container on 80
//       committed inputs (tests/golden/make_reference_py_vectors.py -> reference_py_vectors.json; tests/test_oracle_golden.py).  The Rust
//       path as a whole stays unpinned.
//
// What is restated (all paths relative to /root/reference):
//   src/read.rs:157-174      Read::encode_rle          -> orc::encode_rle
//   src/read.rs:176-211      Read::extract_density     -> orc::extract_density
//   nthash 0.5.1 (external)  NtHashIterator, ntf64/ntr64/ntc64 -> orc::nt*
//   src/kmer_vec.rs:16-43    KmerVec prefix/suffix/reverse/normalize -> orc::kv_*
//   src/main.rs:756-781      k-min-mer window loop     -> Graph::process_read
//   src/main.rs:632-709      add_kminmer (non-Bloom)   -> Graph::add_kminmer
//   src/main.rs:922-929      abundance filter          -> Graph::filter
//   src/main.rs:1014-1117    S/L lines, presimp        -> Graph::emit
//   src/utils.rs:3-24        revcomp                   -> orc::revcomp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

using u8 = uint8_t; using u16 = uint16_t; using u32 = uint32_t; using u64 = uint64_t;
typedef std::vector<u64> Kmer;

enum { ORC_OK = 0, ORC_E_ALPHABET = -1, ORC_E_PARAM = -2 };

// ---------------------------------------------------------------------------------------------
// nthash crate 0.5.1, src/lib.rs: H_LOOKUP / RC_LOOKUP.  Every byte other than A,C,G,T,N maps to
// the sentinel 1, and h()/rc() panic on the sentinel ("Non-ACGTN nucleotide encountered").
// ---------------------------------------------------------------------------------------------
static const u64 SEED_A = 0x3c8bfbb395c60474ULL, SEED_C = 0x3193c18562a02b4cULL,
                 SEED_G = 0x20323ed082572324ULL, SEED_T = 0x295549f54be24456ULL;

static inline bool nt_h(u8 c, u64* out) {
    switch (c) { case 'A': *out = SEED_A; return true; case 'C': *out = SEED_C; return true;
                 case 'G': *out = SEED_G; return true; case 'T': *out = SEED_T; return true;
                 case 'N': *out = 0; return true; default: return false; }
}
static inline bool nt_rc(u8 c, u64* out) {
    switch (c) { case 'A': *out = SEED_T; return true; case 'C': *out = SEED_G; return true;
                 case 'G': *out = SEED_C; return true; case 'T': *out = SEED_A; return true;
                 case 'N': *out = 0; return true; default: return false; }
}
static inline u64 rol(u64 x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
static inline u64 ror(u64 x, unsigned r) { r &= 63; return r ? (x >> r) | (x << (64 - r)) : x; }

// nthash::ntf64(s, i, k): forward hash of s[i..i+k]
static int ntf64(const u8* s, size_t i, size_t k, u64* out) {
    u64 fh = 0, v;
    for (size_t j = 0; j < k; ++j) { if (!nt_h(s[i + j], &v)) return ORC_E_ALPHABET; fh ^= rol(v, (unsigned)(k - j - 1)); }
    *out = fh; return ORC_OK;
}
// nthash::ntr64(s, i, k): reverse-complement hash of s[i..i+k]
static int ntr64(const u8* s, size_t i, size_t k, u64* out) {
    u64 rh = 0, v;
    for (size_t j = 0; j < k; ++j) { if (!nt_rc(s[i + j], &v)) return ORC_E_ALPHABET; rh ^= rol(v, (unsigned)j); }
    *out = rh; return ORC_OK;
}
static int ntc64(const u8* s, size_t i, size_t k, u64* out) {
    u64 f, r; int e;
    if ((e = ntf64(s, i, k, &f))) return e;
    if ((e = ntr64(s, i, k, &r))) return e;
    *out = std::min(f, r); return ORC_OK;
}

// nthash::NtHashIterator::{new,next}: rolling form.  Yields len-k+1 canonical hashes.
// (new() fails with Err when k > len; the reference guards that at read.rs:193.)
static int nthash_iter(const u8* s, size_t n, size_t k, std::vector<u64>& out) {
    out.clear();
    if (k > n || k == 0) return ORC_E_PARAM;
    u64 fh = 0, rh = 0, v;
    for (size_t i = 0; i < k; ++i) { if (!nt_h(s[i], &v)) return ORC_E_ALPHABET; fh ^= rol(v, (unsigned)(k - i - 1)); }
    for (size_t i = 0; i < k; ++i) { size_t j = k - 1 - i; if (!nt_rc(s[j], &v)) return ORC_E_ALPHABET; rh ^= rol(v, (unsigned)(k - i - 1)); }
    size_t max_idx = n - k + 1;
    out.reserve(max_idx);
    for (size_t cur = 0; cur < max_idx; ++cur) {
        if (cur != 0) {
            size_t i = cur - 1; u64 hi, hk, ri, rk;
            if (!nt_h(s[i], &hi) || !nt_h(s[i + k], &hk)) return ORC_E_ALPHABET;
            if (!nt_rc(s[i], &ri) || !nt_rc(s[i + k], &rk)) return ORC_E_ALPHABET;
            fh = rol(fh, 1) ^ rol(hi, (unsigned)k) ^ hk;
            rh = ror(rh, 1) ^ ror(ri, 1) ^ rol(rk, (unsigned)(k - 1));
        }
        out.push_back(std::min(rh, fh));
    }
    return ORC_OK;
}

// src/read.rs:183 — `((density as f64) * (u64::max_value() as f64)) as u64`.
// u64::MAX as f64 rounds to 2^64; Rust's float->int `as` truncates toward zero and saturates.
static u64 hash_bound(double density) {
    double v = density * 18446744073709551616.0;
    if (!(v > 0.0)) return 0;                       // NaN and negatives -> 0
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (u64)v;
}

// src/read.rs:157-174 — homopolymer compression.  A byte is dropped iff it equals the previous
// byte AND is one of "ACTGactgNn".  pos_vec[j] = raw index of the first byte of run j.
// Empty input yields ("#", [0]).
static void encode_rle(const u8* s, size_t n, std::string& hpc, std::vector<u64>& pos) {
    hpc.clear(); pos.clear();
    int prev = '#'; u64 prev_i = 0;
    for (size_t i = 0; i < n; ++i) {
        int c = s[i];
        if (c == prev && std::strchr("ACTGactgNn", c) != nullptr && c != 0) continue;
        if (prev != '#') { hpc.push_back((char)prev); pos.push_back(prev_i); prev_i = i; }
        prev = c;
    }
    hpc.push_back((char)prev); pos.push_back(prev_i);
}

static std::string revcomp(const std::string& s);

// --lmer-counts (robust minimizers).  src/main.rs:544-566: every line "lmer count" of the counts file goes into a map keyed by
// min(lmer, revcomp(lmer)) (later lines overwrite).  src/minimizers.rs:53-113 (minimizers_preparation, the `lmer_counts.len() > 0`
// branch): an l-mer of the map is SELECTED iff it is not skipped (count >= lmer_counts_max or count <= lmer_counts_min, :80-83) and
// `hash as f64 / u64::MAX as f64 <= density` in f64 (:91-98); both it and its reverse complement then map to its canonical ntHash
// (:99-106).  `Some` map = the mode is on (params.has_lmer_counts).
typedef std::unordered_map<std::string, u64> LmerMap;
static int lmer_map_build(const std::vector<std::pair<std::string, u32>>& lines, size_t l, double density, u32 cmin, u32 cmax, LmerMap& out) {
    std::unordered_map<std::string, u32> counts;                                    // main.rs:544
    for (const auto& ln : lines) {
        const std::string rc = revcomp(ln.first);
        counts[ln.first < rc ? ln.first : rc] = ln.second;                          // main.rs:560-564
    }
    out.clear();
    for (const auto& kv : counts) {
        const std::string& lmer = kv.first;                                         // min(x, revcomp(x)) of a canonical key is the key (minimizers.rs:64-65)
        if (lmer.size() < l) return ORC_E_PARAM;                                    // ntc64 would index past the string
        const bool skip = kv.second >= cmax || kv.second <= cmin;                   // minimizers.rs:80
        u64 h; int e = ntc64((const u8*)lmer.data(), 0, l, &h); if (e) return e;    // :88
        double hn = (double)h / 18446744073709551616.0;                             // :89-90 (u64::MAX as f64 == 2^64)
        if (skip) hn = 1.0;                                                         // :91-95
        if (hn <= density) { out[lmer] = h; out[revcomp(lmer)] = h; }               // :96-106
    }
    return ORC_OK;
}

// src/read.rs:176-211 — density sketch; lmer_map != nullptr: the --lmer-counts branch :200-205 (an l-mer that is not in the map is
// skipped, the stored hash replaces the computed one — it is the same value).
struct Sketch { std::vector<u64> transformed; std::vector<u64> pos; };
static int extract_density(const u8* s, size_t n, size_t l, double density, bool already_hpc, Sketch& out, const LmerMap* lmer_map = nullptr) {
    out.transformed.clear(); out.pos.clear();
    u64 bound = hash_bound(density);
    std::string hpc; std::vector<u64> posvec;
    const u8* seq; size_t len;
    if (!already_hpc) { encode_rle(s, n, hpc, posvec); seq = (const u8*)hpc.data(); len = hpc.size(); }
    else { seq = s; len = n; }
    if (len < l) return ORC_OK;                                  // read.rs:193-195
    std::vector<u64> hs;
    int e = nthash_iter(seq, len, l, hs);                        // read.rs:196 (.unwrap(): panics on bad byte)
    if (e) return e;
    for (size_t i = 0; i < hs.size(); ++i) {
        if (hs[i] <= bound) {                                    // inclusive, read.rs:196
            u64 h = hs[i];
            if (lmer_map) {                                      // read.rs:200-205
                auto it = lmer_map->find(std::string((const char*)seq + i, l));
                if (it == lmer_map->end()) continue;
                h = it->second;
            }
            out.pos.push_back(already_hpc ? (u64)i : posvec[i]); // read.rs:206-207
            out.transformed.push_back(h);
        }
    }
    return ORC_OK;
}

// src/kmer_vec.rs:16-43
static Kmer kv_suffix(const Kmer& a) { return Kmer(a.begin() + 1, a.end()); }
static Kmer kv_prefix(const Kmer& a) { return Kmer(a.begin(), a.end() - 1); }
static Kmer kv_reverse(const Kmer& a) { return Kmer(a.rbegin(), a.rend()); }
static std::pair<Kmer, bool> kv_normalize(const Kmer& a) {
    Kmer rev = kv_reverse(a);
    if (a < rev) return {a, false};     // lexicographic Vec<u64> order (kmer_vec.rs:73-77)
    return {rev, true};                 // palindrome -> reversed = true
}

// src/utils.rs:3-24
static std::string revcomp(const std::string& s) {
    std::string r(s.rbegin(), s.rend());
    for (auto& c : r) {
        switch (c) { case 'a': c = 't'; break; case 'c': c = 'g'; break; case 't': c = 'a'; break; case 'g': c = 'c'; break;
                     case 'u': c = 'a'; break; case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'T': c = 'A'; break;
                     case 'G': c = 'C'; break; case 'U': c = 'A'; break; default: c = 'N'; }
    }
    return r;
}

// src/main.rs:60 — DbgEntry {index: u32, abundance: u16, seqlen: u32, shift: (u16, u16)}
struct Entry { u32 index; u16 abundance; u32 seqlen; u16 shift0, shift1;
               // provenance of the occurrence whose seqlen/shift are stored (not in the reference struct;
               // equals what the .sequences line of main.rs:702 is built from)
               u64 src_read, src_start, src_end, shift_full0, shift_full1; u8 reversed; };

struct SeqLine { u32 index; Kmer key; u64 read, start, end; u8 reversed; u64 s0, s1; };   // main.rs:702
struct Edge { u32 n1; char o1; u32 n2; char o2; u32 overlap; };

// src/read.rs:43-52 — the integer hash of the syncmer scheme (wrapping u64 arithmetic of a release build)
static u64 sync_hash(u64 key, u64 mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}
// src/read.rs:23-39 SEQ_NT4_TABLE: A/a 0, C/c 1, G/g 2, T/t/U/u 3, bytes 0..3 themselves, everything else 4
static int nt4(u8 c) {
    switch (c) {
        case 0: case 'A': case 'a': return 0;
        case 1: case 'C': case 'c': return 1;
        case 2: case 'G': case 'g': return 2;
        case 3: case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}
// src/read.rs:215-352 — Read::extract_syncmers: an l-mer is kept when the smallest of its l-s+1 canonical s-mer hashes sits at
// the middle s-mer (position t-1, t = ceil((l-s+1)/2)) AND hash(canonical l-mer) <= density * 4^l.  The minimum is TRACKED by the
// reference's sliding deque (update_window, read.rs:55-80): leftmost minimum of the first full window, afterwards a new element
// replaces it only when strictly smaller, and when the tracked element leaves, the window is rescanned from the back (rightmost
// minimum).  A byte outside ACGTU (either case) resets everything.  s = 0: every l-mer is a candidate (read.rs:319-333).
static int extract_syncmers(const u8* inp, size_t n, size_t l, size_t sm, double density, bool already_hpc, Sketch& out) {
    out.transformed.clear(); out.pos.clear();
    if (l < 1 || l > 31 || sm > l) return ORC_E_PARAM;
    const u64 hash_bound_l = (u64)(density * (double)(1ull << (2 * l)));     // read.rs:218 (density as f64 * 4^l as f64) as u64, saturating
    const double vb = density * (double)(1ull << (2 * l));
    const u64 bound = !(vb > 0.0) ? 0 : (vb >= 18446744073709551616.0 ? UINT64_MAX : hash_bound_l);
    std::string hpc; std::vector<u64> posvec;
    const u8* seq; size_t len;
    if (!already_hpc) { encode_rle(inp, n, hpc, posvec); seq = (const u8*)hpc.data(); len = hpc.size(); }
    else { seq = inp; len = n; }
    if (len < l) return ORC_OK;
    const u64 smask = sm ? ((1ull << 2 * sm) - 1) : 0, lmask = (1ull << 2 * l) - 1;
    const size_t t = (l - sm + 1 + 1) / 2;
    std::vector<u64> qs; std::vector<size_t> qs_pos;        // the deque (front = index 0)
    size_t qs_size = 0; u64 min_val = UINT64_MAX; long min_pos = -1;
    u64 xl[2] = {0, 0}, xs[2] = {0, 0}; size_t lp = 0;
    const u64 lshift = (l - 1) * 2, sshift = sm ? (sm - 1) * 2 : 0;
    auto emit = [&](size_t i) {
        const u64 yl = xl[0] < xl[1] ? xl[0] : xl[1];
        const u64 h = sync_hash(yl, lmask);
        if (h <= bound) { out.transformed.push_back(h); out.pos.push_back(already_hpc ? (u64)(i - l + 1) : posvec[i - l + 1]); }
    };
    for (size_t i = 0; i < len; ++i) {
        const int c = nt4(seq[i]);
        if (c < 4) {
            xl[0] = (xl[0] << 2 | (u64)c) & lmask;
            xl[1] = xl[1] >> 2 | (u64)(3 - c) << lshift;
            if (sm) { xs[0] = (xs[0] << 2 | (u64)c) & smask; xs[1] = xs[1] >> 2 | (u64)(3 - c) << sshift; }
            ++lp;
            if (sm != 0) {
                if (lp >= sm) {
                    const u64 ys = xs[0] < xs[1] ? xs[0] : xs[1];
                    const u64 hs = sync_hash(ys, smask);
                    if (qs_size < l - sm) { qs.push_back(hs); qs_pos.push_back(i - sm + 1); ++qs_size; }
                    else if (qs_size == l - sm) {
                        qs.push_back(hs); qs_pos.push_back(i - sm + 1); ++qs_size;
                        for (size_t j = 0; j < qs_size; ++j) if (qs[j] < min_val) { min_val = qs[j]; min_pos = (long)qs_pos[j]; }
                        if (min_pos == (long)qs_pos[t - 1]) emit(i);
                    } else {
                        // update_window (read.rs:55-80)
                        const size_t popped = qs_pos.front();
                        qs.erase(qs.begin()); qs_pos.erase(qs_pos.begin());
                        qs.push_back(hs); qs_pos.push_back(i - sm + 1);
                        if (min_pos == (long)popped) {
                            min_val = UINT64_MAX; min_pos = (long)(i - sm + 1);
                            for (size_t j = qs.size(); j-- > 0;) if (qs[j] < min_val) { min_val = qs[j]; min_pos = (long)qs_pos[j]; }
                        } else if (hs < min_val) { min_val = hs; min_pos = (long)(i - sm + 1); }
                        if (min_pos == (long)qs_pos[t - 1]) emit(i);
                    }
                }
            } else if (lp >= l) emit(i);
        } else {
            min_val = UINT64_MAX; min_pos = -1; lp = 0; xs[0] = xs[1] = xl[0] = xl[1] = 0; qs_size = 0; qs.clear(); qs_pos.clear();
        }
    }
    return ORC_OK;
}

struct Graph {
    size_t k, l; double density; u16 minabund; bool already_hpc; float presimp;
    bool syncmers = false; size_t sync_s = 0;      // --syncmers / -s (src/read.rs:88)
    bool has_lmer_counts = false; LmerMap lmer_map; // --lmer-counts (src/main.rs:499-503, 571-575)
    std::map<Kmer, Entry> nodes;                  // dbg_nodes, main.rs:595 (ordered map: deterministic, same contents)
    u64 node_index = 0;                           // NODE_INDEX, main.rs:598
    std::vector<SeqLine> seqlines;
    u64 n_reads = 0, n_minimizers = 0, n_windows = 0, n_nodes_before = 0, presimp_removed = 0;
    std::vector<Edge> edges;
    std::vector<const std::pair<const Kmer, Entry>*> order;    // nodes sorted by index after filter

    // src/main.rs:632-709, branch `else` of :636 (no Bloom filter), called with seq=None,
    // read_seq=Some, read_offsets=Some((start, end, seqlen)).
    void add_kminmer(const Kmer& node, bool seq_reversed, u64 s0, u64 s1, u64 read, u64 start, u64 end, u64 seqlen_usize) {
        u16 previous_abundance;
        auto it = nodes.find(node);
        if (it == nodes.end()) {                                              // :657-670
            Entry e{}; e.index = (u32)(node_index++); e.abundance = 0; e.seqlen = (u32)seqlen_usize;
            e.shift0 = (u16)s0; e.shift1 = (u16)s1;
            e.src_read = read; e.src_start = start; e.src_end = end; e.shift_full0 = s0; e.shift_full1 = s1; e.reversed = seq_reversed;
            it = nodes.emplace(node, e).first;
        }
        Entry& em = it->second;                                               // :676-686
        previous_abundance = em.abundance;
        if (previous_abundance == (u16)(minabund - 1)) {
            em.seqlen = (u32)seqlen_usize; em.shift0 = (u16)s0; em.shift1 = (u16)s1;
            em.src_read = read; em.src_start = start; em.src_end = end; em.shift_full0 = s0; em.shift_full1 = s1; em.reversed = seq_reversed;
        }
        em.abundance = (u16)(em.abundance + 1);                               // u16, wraps in release builds
        if (previous_abundance >= 1 || minabund == 1) {                       // :693 (params.reference is false on this path)
            if (previous_abundance == (u16)(minabund - 1))                    // :696
                seqlines.push_back(SeqLine{em.index, node, read, start, end, (u8)seq_reversed, s0, s1});
        }
    }

    // src/main.rs:730-785 (process_read_aux), window loop :756-781
    int process_read(const u8* s, size_t n, u64 read_ordinal) {
        Sketch sk; int e = syncmers ? extract_syncmers(s, n, l, sync_s, density, already_hpc, sk)
                                    : extract_density(s, n, l, density, already_hpc, sk, has_lmer_counts ? &lmer_map : nullptr);
        if (e) return e;
        ++n_reads; n_minimizers += sk.transformed.size();
        const auto& T = sk.transformed; const auto& P = sk.pos;
        if (T.size() > k) {                                                   // STRICT '>' (:756)
            for (size_t i = 0; i < T.size() - k + 1; ++i) {
                Kmer node(T.begin() + i, T.begin() + i + k);                  // make_from
                auto [norm, reversed] = kv_normalize(node);
                u64 second = reversed ? P[i + k - 1] - P[i + k - 2] : P[i + 1] - P[i];          // :769-772
                u64 second_to_last = reversed ? P[i + 1] - P[i] : P[i + k - 1] - P[i + k - 2];  // :773-776
                u64 start = P[i], end = P[i + k - 1] + l, seqlen = P[i + k - 1] + 1 - P[i] + 1; // :778
                add_kminmer(norm, reversed, second, second_to_last, read_ordinal, start, end, seqlen);
                ++n_windows;
            }
        }
        return ORC_OK;
    }

    // src/main.rs:922-929
    void filter() {
        n_nodes_before = nodes.size();
        if (minabund > 1)
            for (auto it = nodes.begin(); it != nodes.end();) { if (it->second.abundance < minabund) it = nodes.erase(it); else ++it; }
        order.clear();
        for (auto& kv : nodes) order.push_back(&kv);
        std::sort(order.begin(), order.end(), [](auto a, auto b) { return a->second.index < b->second.index; });
    }

    // src/main.rs:1014-1117 (edges; the S-lines are just the node table)
    void emit() {
        std::map<Kmer, std::vector<const std::pair<const Kmer, Entry>*>> km_index;
        for (auto p : order) {                                                // :1017-1033
            km_index[kv_normalize(kv_prefix(p->first)).first].push_back(p);
            km_index[kv_normalize(kv_suffix(p->first)).first].push_back(p);
        }
        std::set<std::pair<u32, u32>> removed;
        std::vector<Edge> vec_edges;
        for (auto p1 : order) {                                               // :1041
            const Kmer& n1 = p1->first; const Entry& e1 = p1->second;
            Kmer rev_n1 = kv_reverse(n1);
            Kmer keys[2] = {kv_normalize(kv_suffix(n1)).first, kv_normalize(kv_prefix(n1)).first};
            for (auto& key : keys) {
                auto f = km_index.find(key);
                if (f == km_index.end()) continue;
                struct Pot { const Entry* e; char o1, o2; };
                std::vector<Pot> pot;
                for (auto p2 : f->second) {
                    const Kmer& n2 = p2->first; const Entry* e2 = &p2->second;
                    Kmer rev_n2 = kv_reverse(n2);
                    if (kv_suffix(n1) == kv_prefix(n2)) pot.push_back({e2, '+', '+'});
                    if (kv_suffix(n1) == kv_prefix(rev_n2)) pot.push_back({e2, '+', '-'});
                    if (kv_suffix(rev_n1) == kv_prefix(n2)) pot.push_back({e2, '-', '+'});
                    if (kv_suffix(rev_n1) == kv_prefix(rev_n2)) pot.push_back({e2, '-', '-'});
                }
                if (pot.empty()) continue;
                u16 amax = 0; for (auto& q : pot) amax = std::max(amax, q.e->abundance);
                u16 aref = std::min(amax, e1.abundance);
                for (auto& q : pot) {
                    if (presimp > 0.0f && pot.size() >= 2 && (float)q.e->abundance < presimp * (float)aref) {   // :1083
                        ++presimp_removed; removed.insert({e1.index, q.e->index}); continue;
                    }
                    u16 shift = q.o1 == '+' ? e1.shift0 : e1.shift1;
                    u32 ov = std::min((u32)(e1.seqlen - (u32)shift), (u32)(q.e->seqlen - 1u));   // :1091 (u32 wrapping)
                    vec_edges.push_back({e1.index, q.o1, q.e->index, q.o2, ov});
                }
            }
        }
        edges.clear();
        for (auto& ed : vec_edges) {                                          // :1104-1115 (presimp == 0 writes immediately: same list)
            if (presimp > 0.0f && (removed.count({ed.n1, ed.n2}) || removed.count({ed.n2, ed.n1}))) continue;
            edges.push_back(ed);
        }
    }
};

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C interface for ctypes (tests / bench cpu_baseline only)
// ---------------------------------------------------------------------------------------------
using namespace orc;
extern "C" {

uint64_t orc_hash_bound(double d) { return hash_bound(d); }
int orc_ntf64(const uint8_t* s, uint64_t i, uint64_t k, uint64_t* out) { return ntf64(s, i, k, out); }
int orc_ntr64(const uint8_t* s, uint64_t i, uint64_t k, uint64_t* out) { return ntr64(s, i, k, out); }
int orc_ntc64(const uint8_t* s, uint64_t i, uint64_t k, uint64_t* out) { return ntc64(s, i, k, out); }
// out must hold n-k+1 values; returns count or negative error
int64_t orc_nthash_iter(const uint8_t* s, uint64_t n, uint64_t k, uint64_t* out) {
    std::vector<u64> v; int e = nthash_iter(s, n, k, v); if (e) return e;
    std::memcpy(out, v.data(), v.size() * 8); return (int64_t)v.size();
}
// hpc must hold n+1 bytes, pos n+1 entries; returns hpc length
uint64_t orc_encode_rle(const uint8_t* s, uint64_t n, uint8_t* hpc, uint64_t* pos) {
    std::string h; std::vector<u64> p; encode_rle(s, n, h, p);
    std::memcpy(hpc, h.data(), h.size()); std::memcpy(pos, p.data(), p.size() * 8); return h.size();
}
void orc_revcomp(const uint8_t* s, uint64_t n, uint8_t* out) {
    std::string r = revcomp(std::string((const char*)s, n)); std::memcpy(out, r.data(), n);
}

// Sketch a batch: returns a handle holding concatenated (hash,pos) and per-read offsets.
struct orc_sketch_t { std::vector<u64> hashes, pos, off; int err; u64 err_read; };
orc_sketch_t* orc_sketch(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t l, double density, int already_hpc) {
    auto* r = new orc_sketch_t(); r->err = 0; r->err_read = 0; r->off.push_back(0);
    for (u64 i = 0; i < n_reads; ++i) {
        Sketch sk; int e = extract_density(bases + offsets[i], offsets[i + 1] - offsets[i], l, density, already_hpc != 0, sk);
        if (e && !r->err) { r->err = e; r->err_read = i; }
        r->hashes.insert(r->hashes.end(), sk.transformed.begin(), sk.transformed.end());
        r->pos.insert(r->pos.end(), sk.pos.begin(), sk.pos.end());
        r->off.push_back(r->hashes.size());
    }
    return r;
}
orc_sketch_t* orc_sketch_syncmers(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t l, uint64_t sm, double density, int already_hpc) {
    auto* r = new orc_sketch_t(); r->err = 0; r->err_read = 0; r->off.push_back(0);
    for (u64 i = 0; i < n_reads; ++i) {
        Sketch sk; int e = extract_syncmers(bases + offsets[i], offsets[i + 1] - offsets[i], l, sm, density, already_hpc != 0, sk);
        if (e && !r->err) { r->err = e; r->err_read = i; }
        r->hashes.insert(r->hashes.end(), sk.transformed.begin(), sk.transformed.end());
        r->pos.insert(r->pos.end(), sk.pos.begin(), sk.pos.end());
        r->off.push_back(r->hashes.size());
    }
    return r;
}
// --lmer-counts: the selected l-mers of a counts file given as its lines (lmer i = bytes[offs[i], offs[i+1]), count[i])
struct orc_lmer_map_t { LmerMap m; int err; std::vector<u8> dump; std::vector<u64> dump_hash; };
orc_lmer_map_t* orc_lmer_map_new(const uint8_t* bytes, const uint64_t* offs, const uint32_t* counts, uint64_t n, uint64_t l, double density, uint32_t cmin, uint32_t cmax) {
    auto* r = new orc_lmer_map_t();
    std::vector<std::pair<std::string, u32>> lines;
    for (u64 i = 0; i < n; ++i) lines.emplace_back(std::string((const char*)bytes + offs[i], offs[i + 1] - offs[i]), counts[i]);
    r->err = lmer_map_build(lines, l, density, cmin, cmax, r->m);
    std::vector<std::pair<std::string, u64>> v(r->m.begin(), r->m.end());
    std::sort(v.begin(), v.end());
    for (const auto& kv : v) {            // keys of another length stay in the map (as in the reference) but can never equal a read's l-mer: not listed
        if (kv.first.size() != l) continue;
        r->dump.insert(r->dump.end(), kv.first.begin(), kv.first.end()); r->dump_hash.push_back(kv.second);
    }
    return r;
}
int orc_lmer_map_err(orc_lmer_map_t* m) { return m->err; }
uint64_t orc_lmer_map_n(orc_lmer_map_t* m) { return m->dump_hash.size(); }
const uint8_t* orc_lmer_map_lmers(orc_lmer_map_t* m) { return m->dump.data(); }          // n * l bytes, sorted
const uint64_t* orc_lmer_map_hashes(orc_lmer_map_t* m) { return m->dump_hash.data(); }
void orc_lmer_map_free(orc_lmer_map_t* m) { delete m; }
orc_sketch_t* orc_sketch_lmer(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t l, double density, int already_hpc, orc_lmer_map_t* map) {
    auto* r = new orc_sketch_t(); r->err = map->err; r->err_read = 0; r->off.push_back(0);
    for (u64 i = 0; i < n_reads; ++i) {
        Sketch sk; int e = extract_density(bases + offsets[i], offsets[i + 1] - offsets[i], l, density, already_hpc != 0, sk, &map->m);
        if (e && !r->err) { r->err = e; r->err_read = i; }
        r->hashes.insert(r->hashes.end(), sk.transformed.begin(), sk.transformed.end());
        r->pos.insert(r->pos.end(), sk.pos.begin(), sk.pos.end());
        r->off.push_back(r->hashes.size());
    }
    return r;
}
int orc_sketch_err(orc_sketch_t* s) { return s->err; }
uint64_t orc_sketch_n(orc_sketch_t* s) { return s->hashes.size(); }
const uint64_t* orc_sketch_hashes(orc_sketch_t* s) { return s->hashes.data(); }
const uint64_t* orc_sketch_pos(orc_sketch_t* s) { return s->pos.data(); }
const uint64_t* orc_sketch_off(orc_sketch_t* s) { return s->off.data(); }
void orc_sketch_free(orc_sketch_t* s) { delete s; }

// Full path: reads -> node table (+ edges).  Sequential = reference with --threads 1, no --bf.
struct orc_graph_t { Graph g; int err; u64 err_read;
                     std::vector<u64> keys, src_read, src_start, src_end, sf; std::vector<u32> index, seqlen; std::vector<u16> abund, shift; std::vector<u8> reversed;
                     std::vector<u32> e_n1, e_n2, e_ov; std::vector<u8> e_o1, e_o2;
                     std::vector<u32> sl_index; std::vector<u64> sl_read, sl_start, sl_end, sl_s; std::vector<u8> sl_rev; };
orc_graph_t* orc_graph_new(uint64_t k, uint64_t l, double density, uint32_t minabund, int already_hpc, float presimp) {
    auto* h = new orc_graph_t(); h->err = 0; h->err_read = 0;
    h->g.k = k; h->g.l = l; h->g.density = density; h->g.minabund = (u16)minabund; h->g.already_hpc = already_hpc != 0; h->g.presimp = presimp;
    if (minabund == 0 || minabund > 65535 || k < 2 || l < 1) h->err = ORC_E_PARAM;
    return h;
}
void orc_graph_set_lmer_map(orc_graph_t* h, orc_lmer_map_t* map) { h->g.has_lmer_counts = true; h->g.lmer_map = map->m; if (map->err) h->err = map->err; }
void orc_graph_set_syncmers(orc_graph_t* h, uint64_t sm) { h->g.syncmers = true; h->g.sync_s = sm; if (h->g.l > 31 || sm > h->g.l) h->err = ORC_E_PARAM; }
int orc_graph_ingest(orc_graph_t* h, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t first_read_ordinal) {
    if (h->err) return h->err;
    for (u64 i = 0; i < n_reads; ++i) {
        int e = h->g.process_read(bases + offsets[i], offsets[i + 1] - offsets[i], first_read_ordinal + i);
        if (e) { h->err = e; h->err_read = first_read_ordinal + i; return e; }
    }
    return ORC_OK;
}
int orc_graph_finalize(orc_graph_t* h, int with_edges) {
    if (h->err) return h->err;
    Graph& g = h->g; g.filter(); if (with_edges) g.emit();
    size_t n = g.order.size(), k = g.k;
    h->keys.resize(n * k); h->index.resize(n); h->abund.resize(n); h->seqlen.resize(n); h->shift.resize(2 * n);
    h->src_read.resize(n); h->src_start.resize(n); h->src_end.resize(n); h->sf.resize(2 * n); h->reversed.resize(n);
    for (size_t i = 0; i < n; ++i) {
        auto p = g.order[i]; std::memcpy(&h->keys[i * k], p->first.data(), k * 8);
        const Entry& e = p->second;
        h->index[i] = e.index; h->abund[i] = e.abundance; h->seqlen[i] = e.seqlen; h->shift[2 * i] = e.shift0; h->shift[2 * i + 1] = e.shift1;
        h->src_read[i] = e.src_read; h->src_start[i] = e.src_start; h->src_end[i] = e.src_end; h->sf[2 * i] = e.shift_full0; h->sf[2 * i + 1] = e.shift_full1; h->reversed[i] = e.reversed;
    }
    for (auto& ed : g.edges) { h->e_n1.push_back(ed.n1); h->e_n2.push_back(ed.n2); h->e_ov.push_back(ed.overlap); h->e_o1.push_back((u8)ed.o1); h->e_o2.push_back((u8)ed.o2); }
    for (auto& s : g.seqlines) { h->sl_index.push_back(s.index); h->sl_read.push_back(s.read); h->sl_start.push_back(s.start); h->sl_end.push_back(s.end);
                                 h->sl_rev.push_back(s.reversed); h->sl_s.push_back(s.s0); h->sl_s.push_back(s.s1); }
    return ORC_OK;
}
uint64_t orc_graph_counter(orc_graph_t* h, int which) {
    Graph& g = h->g;
    switch (which) { case 0: return g.n_reads; case 1: return g.n_minimizers; case 2: return g.n_windows; case 3: return g.n_nodes_before;
                     case 4: return g.order.size(); case 5: return g.edges.size(); case 6: return g.presimp_removed; case 7: return g.seqlines.size();
                     case 8: return h->err_read; default: return 0; }
}
const uint64_t* orc_graph_keys(orc_graph_t* h) { return h->keys.data(); }
const uint32_t* orc_graph_index(orc_graph_t* h) { return h->index.data(); }
const uint16_t* orc_graph_abundance(orc_graph_t* h) { return h->abund.data(); }
const uint32_t* orc_graph_seqlen(orc_graph_t* h) { return h->seqlen.data(); }
const uint16_t* orc_graph_shift(orc_graph_t* h) { return h->shift.data(); }
const uint64_t* orc_graph_shift_full(orc_graph_t* h) { return h->sf.data(); }
const uint64_t* orc_graph_src_read(orc_graph_t* h) { return h->src_read.data(); }
const uint64_t* orc_graph_src_start(orc_graph_t* h) { return h->src_start.data(); }
const uint64_t* orc_graph_src_end(orc_graph_t* h) { return h->src_end.data(); }
const uint8_t* orc_graph_reversed(orc_graph_t* h) { return h->reversed.data(); }
const uint32_t* orc_graph_edge_n1(orc_graph_t* h) { return h->e_n1.data(); }
const uint32_t* orc_graph_edge_n2(orc_graph_t* h) { return h->e_n2.data(); }
const uint32_t* orc_graph_edge_overlap(orc_graph_t* h) { return h->e_ov.data(); }
const uint8_t* orc_graph_edge_o1(orc_graph_t* h) { return h->e_o1.data(); }
const uint8_t* orc_graph_edge_o2(orc_graph_t* h) { return h->e_o2.data(); }
const uint32_t* orc_graph_seqline_index(orc_graph_t* h) { return h->sl_index.data(); }
const uint64_t* orc_graph_seqline_read(orc_graph_t* h) { return h->sl_read.data(); }
const uint64_t* orc_graph_seqline_start(orc_graph_t* h) { return h->sl_start.data(); }
const uint64_t* orc_graph_seqline_end(orc_graph_t* h) { return h->sl_end.data(); }
const uint8_t* orc_graph_seqline_rev(orc_graph_t* h) { return h->sl_rev.data(); }
const uint64_t* orc_graph_seqline_shift(orc_graph_t* h) { return h->sl_s.data(); }
void orc_graph_free(orc_graph_t* h) { delete h; }

// Test hook: run the reference's graph emitter (src/main.rs:1014-1117) on a node table supplied by the caller
// (e.g. the one libmdbg_hip produced).  The emitter is a pure function of {key, index, abundance, seqlen, shift}.
orc_graph_t* orc_graph_from_nodes(uint64_t k, uint64_t n, const uint64_t* keys, const uint32_t* index, const uint16_t* abundance,
                                  const uint32_t* seqlen, const uint16_t* shift, float presimp) {
    auto* h = new orc_graph_t(); h->err = 0; h->err_read = 0;
    h->g.k = k; h->g.l = 0; h->g.density = 0; h->g.minabund = 1; h->g.already_hpc = false; h->g.presimp = presimp;
    for (uint64_t i = 0; i < n; ++i) {
        Entry e{}; e.index = index[i]; e.abundance = abundance[i]; e.seqlen = seqlen[i]; e.shift0 = shift[2 * i]; e.shift1 = shift[2 * i + 1];
        h->g.nodes.emplace(Kmer(keys + i * k, keys + (i + 1) * k), e);
    }
    return h;
}

// Timing-only multi-threaded variant for bench.py's cpu_baseline leg: the same per-read functions,
// one worker per thread over a contiguous slice of reads with a thread-local counting map, merged at
// the end (the reference shares one DashMap between its --threads workers, main.rs:595,834).
// Returns the number of nodes with abundance >= minabund; *n_windows gets the occurrence count.
struct VecHash { size_t operator()(const Kmer& v) const { u64 h = 0x9E3779B97F4A7C15ULL; for (u64 x : v) { h ^= x; h *= 0xff51afd7ed558ccdULL; h ^= h >> 32; } return (size_t)h; } };
// Multi-threaded counting variant (bench.py's cpu_baseline): the reference counts into one concurrent map shared by its worker threads
// (DashMap, src/main.rs:595,834); the port does it in two phases without a shared map — every thread sketches its share of the reads and
// deals the canonical k-min-mers into one bucket per thread by key hash (flat arrays, k values per key), then every thread counts the
// bucket it owns — so no thread waits for another and nothing is merged serially.
struct KeyRef { const u64* p; u32 k; };
struct KeyRefHash { size_t operator()(const KeyRef& a) const { u64 h = 0x9E3779B97F4A7C15ULL; for (u32 i = 0; i < a.k; ++i) { h ^= a.p[i]; h *= 0xff51afd7ed558ccdULL; h ^= h >> 32; } return (size_t)h; } };
struct KeyRefEq { bool operator()(const KeyRef& a, const KeyRef& b) const { return std::memcmp(a.p, b.p, (size_t)a.k * 8) == 0; } };
// Order-free digest of a filtered node table {(key, abundance)} (what dbg_nodes holds after src/main.rs:922-929): per node
//   h = 0x243F6A8885A308D3 ^ abundance;  h = fmix64(h ^ key[j]) for j = 0 .. k-1   (fmix64: the 64-bit finaliser of MurmurHash3)
// and over the table digest[0] = sum of h mod 2^64, digest[1] = XOR of h.  Two tables with the same digest and node count hold the same set of
// (key, abundance) pairs (up to a 2^-64-ish collision); the digests of disjoint partitions add / XOR up to the whole table's.  The library computes the same
// formula over its device table (mdbg_nodes_digest, include/mdbg_hip.h): bench.py compares the two on the whole benchmark workload in every run.
static inline u64 orc_fmix64(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static inline u64 orc_node_hash(const u64* key, u64 k, u16 abundance) {
    u64 h = 0x243F6A8885A308D3ULL ^ (u64)abundance;
    for (u64 j = 0; j < k; ++j) h = orc_fmix64(h ^ key[j]);
    return h;
}
int64_t orc_count_digest_threaded(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t k, uint64_t l, double density,
                                  uint32_t minabund, int already_hpc, int threads, uint64_t* n_windows, uint64_t* digest);
int64_t orc_count_threaded(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t k, uint64_t l, double density,
                           uint32_t minabund, int already_hpc, int threads, uint64_t* n_windows) {
    return orc_count_digest_threaded(bases, offsets, n_reads, k, l, density, minabund, already_hpc, threads, n_windows, nullptr);
}
// digest (null: not wanted): [0] sum, [1] XOR of the solid nodes' hashes (above)
int64_t orc_count_digest_threaded(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t k, uint64_t l, double density,
                                  uint32_t minabund, int already_hpc, int threads, uint64_t* n_windows, uint64_t* digest) {
    if (threads < 1) threads = 1;
    const size_t P = (size_t)threads;
    std::vector<std::vector<std::vector<u64>>> deal(P, std::vector<std::vector<u64>>(P));      // deal[producer][owner]: keys, k values each
    std::vector<u64> wins(P, 0); std::vector<int> errs(P, 0);
    auto produce = [&](size_t t) {
        const u64 lo = n_reads * t / P, hi = n_reads * (t + 1) / P;
        std::vector<u64> key(k);
        for (u64 r = lo; r < hi; ++r) {
            Sketch sk; int e = extract_density(bases + offsets[r], offsets[r + 1] - offsets[r], l, density, already_hpc != 0, sk);
            if (e) { errs[t] = e; return; }
            const auto& T = sk.transformed;
            if (T.size() > k) for (size_t i = 0; i + k <= T.size(); ++i) {
                // KmerVec::normalize (src/kmer_vec.rs:34-39): the smaller of the window and its reverse, ties -> reversed (same values)
                bool rev = true;
                for (size_t j = 0; j < k; ++j) { const u64 a = T[i + j], b = T[i + k - 1 - j]; if (a != b) { rev = !(a < b); break; } }
                for (size_t j = 0; j < k; ++j) key[j] = rev ? T[i + k - 1 - j] : T[i + j];
                const size_t owner = KeyRefHash()(KeyRef{key.data(), (u32)k}) % P;
                auto& dst = deal[t][owner];
                dst.insert(dst.end(), key.begin(), key.end());
                ++wins[t];
            }
        }
    };
    std::vector<int64_t> solids(P, 0);
    std::vector<u64> dsum(P, 0), dxor(P, 0);
    auto count = [&](size_t o) {
        size_t n = 0;
        for (size_t t = 0; t < P; ++t) n += deal[t][o].size() / k;
        std::unordered_map<KeyRef, u32, KeyRefHash, KeyRefEq> m;
        m.reserve(n);
        for (size_t t = 0; t < P; ++t) { const auto& v = deal[t][o]; for (size_t q = 0; q + k <= v.size(); q += k) m[KeyRef{v.data() + q, (u32)k}] += 1; }
        int64_t s = 0; u64 ds = 0, dx = 0;
        for (auto& kv : m) if ((u16)kv.second >= (u16)minabund || minabund <= 1) {
            ++s;
            if (digest) { const u64 h = orc_node_hash(kv.first.p, k, (u16)kv.second); ds += h; dx ^= h; }
        }
        solids[o] = s; dsum[o] = ds; dxor[o] = dx;
    };
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        for (size_t t = 1; t < P; ++t) th.emplace_back(fn, t);
        fn((size_t)0);
        for (auto& x : th) x.join();
    };
    run(produce);
    for (size_t t = 0; t < P; ++t) if (errs[t]) return errs[t];
    run(count);
    int64_t solid = 0; u64 w = 0;
    for (auto x : solids) solid += x;
    for (auto x : wins) w += x;
    if (n_windows) *n_windows = w;
    if (digest) { digest[0] = digest[1] = 0; for (size_t o = 0; o < P; ++o) { digest[0] += dsum[o]; digest[1] ^= dxor[o]; } }
    return solid;
}

}  // extern "C"
