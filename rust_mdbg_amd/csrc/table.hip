// table.hip — k-min-mer windows, canonicalisation and the GPU-resident counting table (gfx950).
//
// Replaces the window loop of process_read_aux (rust-mdbg src/main.rs:756-781), KmerVec::normalize
// (src/kmer_vec.rs:34-39), add_kminmer's counting upsert on DashMap (src/main.rs:632-691, non-Bloom
// branch) and the abundance filter (src/main.rs:922-929).
//
// Table: open addressing, linear probing, 32-byte slots.  A slot is claimed with ONE 64-bit CAS on
//   word = fingerprint(30) | src(1) | rev(1) | rep(32)
// where `rep` points at a representative occurrence of the key (an index into the resident minimizer
// array, or into the routed-record arena when src=1) — the k*8-byte key itself is never copied.  A
// fingerprint hit is confirmed by comparing the full canonical key with the representative's, so the
// table is exact.  The occurrence that CLAIMS a slot costs exactly one atomic (the CAS): its ordinal
// (ordinal = read ordinal << 26 | window index) is recoverable from `rep`.  Every later occurrence of the key adds 1
// to `count` and offers its ordinal to the A smallest kept in m1, m2, mx[]; at finalize the claimer is merged back in
// (slot_view): the smallest ordinal gives DbgEntry.index (order of first sighting), the A-th gives the sighting whose
// seqlen/shift the reference stores, abundance = count + 1.
#include "mdbg_dev.h"

struct __attribute__((aligned(32))) Slot {
    u64 word;      // ~0 = empty
    u64 m1;        // smallest ordinal
    u64 m2;        // second smallest (A >= 2)
    u32 count;
    u32 pad;
};
constexpr u64 EMPTY = ~0ull;
constexpr u64 HMUL = 0x9E3779B97F4A7C15ull;

struct KeySrc {                   // where representative keys live
    const u64* mh;                // resident minimizer hashes (src = 0): key = mh[rep .. rep+k), orientation in the slot word
    const u64* arena;             // routed records (src = 1), k+2 u64 each: key = arena[rep*(k+2) .. +k), already canonical
    u32 k;
};

__device__ inline u64 load_relaxed(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// canonical element j of the key a slot word stands for
__device__ inline u64 rep_elem(const KeySrc& ks, u64 word, u32 j) {
    const u32 rep = (u32)word;
    if (word & (1ull << 33)) return ks.arena[(u64)rep * (ks.k + 2) + j];
    return (word & (1ull << 32)) ? ks.mh[(u64)rep + ks.k - 1 - j] : ks.mh[(u64)rep + j];
}

// KmerVec::normalize (src/kmer_vec.rs:34-39): true = the reversed vector is the canonical one (ties included)
__device__ inline bool window_reversed(const u64* __restrict__ w, u32 k) {
    for (u32 j = 0; j < k / 2 + 1 && j < k; ++j) {
        const u64 a = w[j], b = w[k - 1 - j];
        if (a < b) return false;
        if (a > b) return true;
    }
    return true;
}
// Hash of a canonical key given by an accessor elem(j): four independent multiply-xorshift chains over j mod 4 (the
// chain of one lane would otherwise be k dependent 64-bit multiplies), folded and finished with fmix64.
template <class ElemFn>
__device__ inline u64 key_hash_fn(ElemFn elem, u32 k) {
    u64 h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull, h2 = 0xA4093822299F31D0ull, h3 = 0x082EFA98EC4E6C89ull;
    u32 j = 0;
    for (; j + 4 <= k; j += 4) {
        h0 = (h0 ^ elem(j)) * HMUL;     h0 ^= h0 >> 29;
        h1 = (h1 ^ elem(j + 1)) * HMUL; h1 ^= h1 >> 29;
        h2 = (h2 ^ elem(j + 2)) * HMUL; h2 ^= h2 >> 29;
        h3 = (h3 ^ elem(j + 3)) * HMUL; h3 ^= h3 >> 29;
    }
    if (j < k) { h0 = (h0 ^ elem(j)) * HMUL; h0 ^= h0 >> 29; }
    if (j + 1 < k) { h1 = (h1 ^ elem(j + 1)) * HMUL; h1 ^= h1 >> 29; }
    if (j + 2 < k) { h2 = (h2 ^ elem(j + 2)) * HMUL; h2 ^= h2 >> 29; }
    return fmix64(h0 ^ rol64(h1, 17) ^ rol64(h2, 31) ^ rol64(h3, 47));
}
__device__ inline u64 key_hash_window(const u64* __restrict__ w, u32 k, bool rev) {
    return key_hash_fn([&](u32 j) { return rev ? w[k - 1 - j] : w[j]; }, k);
}
// The same hash of a window that lies in HBM, with the window's smallest value as a by-product: sixteen values per round trip (the loop of key_hash_fn fetches four, waits,
// mixes: nine dependent round trips at k = 35, and the owner check's loop another k) — the per-entry insertion of listed windows spent its time waiting on those chains.
__device__ inline u64 key_hash_window_hbm(const u64* __restrict__ w, u32 k, bool rev, u64& smallest) {
    u64 h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    const u64* const p = rev ? w + (k - 1) : w;
    const long st = rev ? -1 : 1;
    u64 mn = ~0ull;
    for (u32 j0 = 0; j0 < k; j0 += 16) {
        u64 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { const u32 j = j0 + t < k ? j0 + t : k - 1; v[t] = p[st * (long)j]; }
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (j0 + t < k) { h[t & 3] = (h[t & 3] ^ v[t]) * HMUL; h[t & 3] ^= h[t & 3] >> 29; mn = v[t] < mn ? v[t] : mn; }      // (j0 is a multiple of 4: position j feeds chain j mod 4, as key_hash_fn)
    }
    smallest = mn;
    return fmix64(h[0] ^ rol64(h[1], 17) ^ rol64(h[2], 31) ^ rol64(h[3], 47));
}
__device__ inline u64 key_hash_canon(const u64* __restrict__ key, u32 k) {
    return key_hash_fn([&](u32 j) { return key[j]; }, k);
}

struct TableArgs {
    Slot* tab; u64 cap;           // number of slots (any value >= 1024)
    u64* mx;                      // [capacity][A-2] further minima when A > 2, else null
    u32 A;
    u64* n_distinct;              // device counter
    KeySrc ks;
    u32 own_world, own_rank;      // replicated-sketch mode: insert only windows owned by own_rank (own_world <= 1: all)
    const u64* own_thr;           // see OwnerSpec
    u32* probe_err;               // set when a probe sequence visited every slot: the table was sized from a wrong window count
    u64* own_inserted;            // sharded counter: owned windows actually inserted (checked against the senders' counts)
    u64 fp_mask;                  // 0x3FFFFFFF; MDBG_WEAK_FP (test hook) leaves two bits, so that most probes meet ANOTHER key behind their fingerprint
    unsigned long long* link_ctr; // non-null (MDBG_COUNT_LINKS): matches confirmed as links are counted here
    u32 no_chain;                 // 1: every fingerprint hit is confirmed by the full comparison (MDBG_NO_CHAIN; see insert_windows_kernel)
    u8* claim;                    // non-null: claim[i] <- 1 when the window starting at minimizer index i CLAIMED its slot (created the key), else 0 — written for every
                                  // index of the span, so the map needs no zeroing; finalize starts from it instead of marking every key's first sighting (fin_mark_kernel)
};

// Home slot of a key hash: range reduction by multiplication, so the capacity need not be a power of two.  It is fed
// from the LOW 40 bits of the hash: the high bits choose the owning rank in the routed mode (route.hip) and the
// fingerprint, and must stay independent of the position inside one rank's table.
__device__ inline u64 home_slot(u64 h, u64 cap) { return __umul64hi(h << 24, cap); }

// insert ordinal x into the slot's A smallest
__device__ inline void push_ordinal(const TableArgs& T, u64 s, u64 x) {
    Slot* e = T.tab + s;
    const u32 A = T.A;
    // cheap reject: x larger than the current A-th smallest (values only ever decrease)
    const u64* last = A == 1 ? &e->m1 : A == 2 ? &e->m2 : &T.mx[s * (A - 2) + (A - 3)];
    if (x > load_relaxed(last)) return;
    u64 carry = x;
    for (u32 lvl = 0; lvl < A && carry != EMPTY; ++lvl) {
        u64* m = lvl == 0 ? &e->m1 : lvl == 1 ? &e->m2 : &T.mx[s * (A - 2) + (lvl - 2)];
        const u64 old = atomicMin((unsigned long long*)m, (unsigned long long)carry);
        if (old > carry) carry = old;          // I displaced `old`; it moves one level down
    }
}

// find-or-claim the slot of a key; same_key(word) = full comparison of my key with the representative a slot word names.
// upsert_slot_from: the same walk entered at slot s after `probes` slots have been looked at already (insert_windows_kernel's second stage).
template <class EqFn>
__device__ inline u64 upsert_slot_from(const TableArgs& T, u64 h, u64 myword_lo, EqFn same_key, bool& claimed, u64 s, u64 probes);
template <class EqFn>
__device__ inline u64 upsert_slot(const TableArgs& T, u64 h, u64 myword_lo, EqFn same_key, bool& claimed) {
    return upsert_slot_from(T, h, myword_lo, same_key, claimed, home_slot(h, T.cap), 0);
}
template <class EqFn>
__device__ inline u64 upsert_slot_from(const TableArgs& T, u64 h, u64 myword_lo, EqFn same_key, bool& claimed, u64 s, u64 probes) {
    claimed = false;
    const u64 fp = (h >> 34) & T.fp_mask;
    const u64 myword = (fp << 34) | myword_lo;
    for (; probes <= T.cap; ++probes) {
        u64 w = load_relaxed(&T.tab[s].word);      // (claiming without looking first — one round trip for a new key instead of two — is slower:
        if (w == EMPTY) {                          //  0.72 instead of 0.65 ms per 6.6 M windows; the kernel is bound by the rate of its atomics)
            const u64 old = atomicCAS((unsigned long long*)&T.tab[s].word, (unsigned long long)EMPTY, (unsigned long long)myword);
            if (old == EMPTY) { wave_agg_inc(T.n_distinct); claimed = true; return s; }
            w = old;
        }
        if ((w >> 34) == fp && same_key(w)) return s;
        s = s + 1 == T.cap ? 0 : s + 1;
    }
    *T.probe_err = 1;                          // table full: cannot happen when it was sized from the true number of windows
    return ~0ull;
}
// comparison of a key given as a contiguous run wl[0..k) (a window staged in LDS with orientation rev_mine, or a routed
// record's canonical key) with the representative in HBM: both are
// contiguous, so the representative is read 16 bytes per load, eight values per round trip; the last round overlaps.
typedef u64 u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ inline bool same_key_window(const KeySrc& ks, u64 w, const u64* wl, bool rev_mine) {
    const u32 k = ks.k, rep = (u32)w;
    const u64* rp; bool cross;                // cross: rp[j] pairs with wl[k-1-j]
    if (w & (1ull << 33)) { rp = ks.arena + (u64)rep * (k + 2); cross = rev_mine; }
    else { rp = ks.mh + rep; cross = rev_mine != ((w & (1ull << 32)) != 0); }
    if (k < 8) {
        u64 diff = 0;
        for (u32 j = 0; j < k; ++j) diff |= rp[j] ^ wl[cross ? k - 1 - j : j];
        return !diff;
    }
    if (k >= 16) {
        // sixteen values per round trip (eight 16-byte loads in flight): k = 35 takes three dependent rounds instead of five
        for (u32 j = 0;; j += 16) {
            if (j + 16 > k) j = k - 16;
            u64x2_a8 a[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) a[t] = *(const u64x2_a8*)(rp + j + 2 * t);
            u64 diff = 0;
            if (!cross) {
                const u64* m = wl + j;
#pragma unroll
                for (int t = 0; t < 8; ++t) diff |= (a[t].x ^ m[2 * t]) | (a[t].y ^ m[2 * t + 1]);
            } else {
                const u64* m = wl + (k - 16 - j);      // m[15-t] pairs with rp[j+t]
#pragma unroll
                for (int t = 0; t < 8; ++t) diff |= (a[t].x ^ m[15 - 2 * t]) | (a[t].y ^ m[14 - 2 * t]);
            }
            if (diff) return false;
            if (j + 16 >= k) return true;
        }
    }
    for (u32 j = 0;; j += 8) {
        if (j + 8 > k) j = k - 8;
        const u64x2_a8 a0 = *(const u64x2_a8*)(rp + j), a1 = *(const u64x2_a8*)(rp + j + 2),
                       a2 = *(const u64x2_a8*)(rp + j + 4), a3 = *(const u64x2_a8*)(rp + j + 6);
        u64 diff;
        if (!cross) {
            const u64* m = wl + j;
            diff = (a0.x ^ m[0]) | (a0.y ^ m[1]) | (a1.x ^ m[2]) | (a1.y ^ m[3]) | (a2.x ^ m[4]) | (a2.y ^ m[5]) | (a3.x ^ m[6]) | (a3.y ^ m[7]);
        } else {
            const u64* m = wl + (k - 8 - j);   // m[7-t] pairs with rp[j+t]
            diff = (a0.x ^ m[7]) | (a0.y ^ m[6]) | (a1.x ^ m[5]) | (a1.y ^ m[4]) | (a2.x ^ m[3]) | (a2.y ^ m[2]) | (a3.x ^ m[1]) | (a3.y ^ m[0]);
        }
        if (diff) return false;
        if (j + 8 >= k) return true;
    }
}

// Owner of a k-min-mer in the replicated-sketch multi-GPU mode: a function of the SMALLEST of its k minimizer hashes (the same for the key
// and its reverse).  Consecutive windows of a read share their smallest hash for (k + 1) / 2 steps on average, so a rank's windows come in
// runs: a run of r windows needs r + k - 1 hashes of the read and nothing else of it, which is what the sketching rank ships to the owner
// (mdbg_dist, "segments") — a few hashes per window instead of the whole sketch to every rank.  (Round 2 hashed both ends and the middle: an
// O(1) function, but neighbouring windows went to unrelated ranks and every rank needed every hash.)
// The smallest of k values that are uniform on [0, bound] (the selected minimizers' hashes) has the distribution function
// 1 - (1 - v / bound)^k, and a minimizer that is small is the smallest of MANY windows: hashing the value to a rank left one of eight
// ranks with 43 % more nodes than the mean (l = 12: the few hundred smallest l-mer hashes carry most windows).  Cutting [0, 1) of that
// distribution function into `world` equal parts gives every rank the same expected share whatever the weights: 1.05 instead of 1.43.
// Owner parameters in device memory (u64 units): [0] = bin multiplier (0: no table), [1 .. 64) thresholds, ascending: rank r owns the minima v with
// thr[r - 1] <= v < thr[r]; from [64]: OWNER_BINS bytes, the owner of every bin of the value range (bin = mulhi64(v, multiplier)) — a MEASURED
// assignment: the multi-GPU layer counts the window minima of its first round per bin, sums the counts over the ranks and deals the bins out
// heaviest first to the least loaded rank (dist_api.inc, build_owner_table), so that a handful of very frequent small hashes (l = 12: ~700 distinct
// homopolymer-compressed 12-mers under the threshold, the smallest one the minimum of 5 % of all windows) cannot leave one rank with 20 % more
// than its share.  Neighbouring windows still share their owner (it is still a function of the smallest hash alone).
constexpr u32 OWNER_BINS = 65536, OWNER_THR_AT = 1, OWNER_TAB_AT = 64, OWNER_PARAM_WORDS = OWNER_TAB_AT + OWNER_BINS / 8;
struct OwnerSpec { u32 world; const u64* thr; };      // thr: the block above (null: hash the value)
__device__ inline u32 owner_of_min(u64 m, u32 k, OwnerSpec os);
__device__ inline u32 window_owner(const u64* __restrict__ w, u32 k, OwnerSpec os) {
    if (os.world <= 1) return 0;
    u64 m = w[0];
    for (u32 j = 1; j < k; ++j) { const u64 x = w[j]; m = x < m ? x : m; }
    return owner_of_min(m, k, os);
}
__device__ inline u32 owner_of_min(u64 m, u32 k, OwnerSpec os) {
    (void)k;
    if (os.world <= 1) return 0;
    if (!os.thr) return (u32)__umul64hi(fmix64(m), (u64)os.world);
    const u64 mul = os.thr[0];
    if (mul) {
        const u64 bin = __umul64hi(m, mul);
        const u32 o = ((const u8*)(os.thr + OWNER_TAB_AT))[bin < OWNER_BINS ? bin : OWNER_BINS - 1];
        return o < os.world ? o : os.world - 1;
    }
    const u64* const thr = os.thr + OWNER_THR_AT;
    u32 lo = 0, hi = os.world - 1;                  // number of thresholds <= m
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (thr[mid] <= m) lo = mid + 1; else hi = mid; }
    return lo;
}
// thr[r - 1] = bound * (1 - (1 - r / world)^(1 / k)), r = 1 .. world - 1: the values at which the distribution function of the window minimum,
// 1 - (1 - v / bound)^k, passes r / world.  Computed ONCE per (k, world, bound) on the device (every rank runs the same code on the same
// hardware: identical thresholds without any host floating point), the windows are then placed by integer comparisons.
__global__ void owner_thresholds_kernel(double bound, u32 k, u32 world, u64* __restrict__ thr) {
    const u32 r = threadIdx.x + 1;
    if (threadIdx.x == 0) thr[0] = 0;               // no measured table (yet)
    thr += OWNER_THR_AT;
    if (r >= world) return;
    const double x = -expm1(log1p(-(double)r / (double)world) / (double)k);
    double v = x * bound;
    thr[r - 1] = v >= 18446744073709549568.0 ? ~0ull : (u64)v;
}
void launch_owner_thresholds(double bound, u32 k, u32 world, u64* thr, hipStream_t s) {
    if (world > 1) hipLaunchKernelGGL(owner_thresholds_kernel, dim3(1), dim3(64), 0, s, bound, k, world, thr);
}

// ---- owner codes --------------------------------------------------------------------------------------------------------------------------------------------
// A window's owner is a function of the SMALLEST of its k hashes, and both owner functions with parameters (OwnerSpec::thr) are monotone in that value up to a final table
// lookup: bin(v) = min(mulhi64(v, mul), OWNER_BINS - 1) for the measured table, rank(v) = number of thresholds <= v without one.  So
// min over the window of code(v) = code(min over the window of v), and the sliding minimum runs on 16-bit CODES: the values are turned into codes once, the doubling rounds
// (min over 2, 4, ... p <= k codes; a window of k is two overlapping stretches of p) move 2 bytes per element instead of 8 — round 3 ran them on the u64 hashes in 33 KB
// of LDS per workgroup, and a kernel that also needs the hashes themselves (the insertion) could not afford them at all.  (No parameters — the value is hashed to a rank —
// is not monotone: callers fall back to window_owner.)
struct OwnerCodes { u64 mul; const u64* thr; u32 world; };      // mul != 0: bins of the measured table; else threshold ranks
__device__ inline OwnerCodes owner_codes_of(OwnerSpec os) { OwnerCodes c; c.thr = os.thr; c.world = os.world; c.mul = os.thr ? os.thr[0] : 0; return c; }
__device__ inline u16 owner_code(u64 v, const OwnerCodes& c) {
    if (c.mul) { const u64 bin = __umul64hi(v, c.mul); return (u16)(bin < OWNER_BINS ? bin : OWNER_BINS - 1); }
    const u64* const thr = c.thr + OWNER_THR_AT;
    u32 lo = 0, hi = c.world - 1;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (thr[mid] <= v) lo = mid + 1; else hi = mid; }
    return (u16)lo;
}
__device__ inline u32 owner_of_code(u16 code, const OwnerCodes& c) {
    if (!c.mul) return code;
    const u32 o = ((const u8*)(c.thr + OWNER_TAB_AT))[code];
    return o < c.world ? o : c.world - 1;
}
// a, b: two LDS arrays of nv u16; value(t): the hash at span position t, valid(t): it exists.  Returns M with M[t] = smallest code among the p values from t on (p = the
// largest power of two <= k): the code of the window starting at span position li is min(M[li], M[li + k - p]).  All 256 threads call it.
template <class ValueFn, class ValidFn>
__device__ inline const u16* span_min_codes(ValueFn value, ValidFn valid, u32 nv, u32 k, const OwnerCodes& c, u16* a, u16* b, u32& p) {
    for (u32 t = threadIdx.x; t < nv; t += 256) a[t] = valid(t) ? owner_code(value(t), c) : (u16)0xFFFFu;
    __syncthreads();
    for (p = 1; 2 * p <= k; p *= 2) {
        for (u32 t = threadIdx.x; t < nv; t += 256) { const u16 x = a[t], y = t + p < nv ? a[t + p] : (u16)0xFFFFu; b[t] = x < y ? x : y; }
        __syncthreads();
        u16* const sw = a; a = b; b = sw;
    }
    return a;
}

// find-or-claim for a whole wave: every lane calls it together, act = the lane has a window (its k values at w, window start = store index i, position li in a
// list that is in window order where consecutive entries are consecutive windows of a read).  claimed: the lane created the key; found: it met its key in slot s.
// The walk of upsert_slot is cut in two.  (1) Every lane walks to the first slot that is empty (claims it) or carries its fingerprint.  (2) The fingerprint
// hits are confirmed: with all lanes' first stage behind them, the loads of the representatives are issued side by side instead of each at the end of its
// lane's own chain of probes — 0.713 -> 0.665 ms per 6.6 M windows, 36.6 -> 33.5 ms for the human data set (profiles/r05_x_insert_two_stage.txt).
// A lane whose hit CONTINUES its left neighbour's — the neighbour (window i - 1) sits in a slot whose representative is store index r, this lane's slot names
// r + 1 (r - 1 when the orientations cross), same orientation relation — is a LINK: k - 1 of its k comparisons are the neighbour's, it compares its newest
// value only (8 bytes instead of 8 k in three cache lines); it is confirmed when it and every lane between it and its HEAD (the nearest lane below that is not a
// link; lane 0 always is one, nothing crosses a wave) passed.  Nothing is taken on trust: every accepted match is a full comparison or follows from one.  Links
// need the neighbouring keys to have been created by neighbouring windows of ONE earlier read: common when a batch holds a copy or two of a region (a file
// streamed in 256-Mbase batches), rare when one launch inserts dozens of copies that race for the claim (the benchmark's 50x batches: no gain there; an
// experiment that compared 16 bytes of every representative — not exact, never shipped: profiles/r05_x_shortcmp.txt — bounds what links can give at -18 % / -22 %).
// A lane that fails either way walks on with full comparisons like upsert_slot.
__device__ inline u64 upsert_wave_h(const TableArgs& T, bool act, u32 li, u64 i, const u64* w, u32 k, bool rev, u64 h, bool& claimed, bool& found);
__device__ inline u64 upsert_wave(const TableArgs& T, bool act, u32 li, u64 i, const u64* w, u32 k, bool& claimed, bool& found) {
    bool rev = false; u64 h = 0;
    if (act) { rev = window_reversed(w, k); h = key_hash_window(w, k, rev); }
    return upsert_wave_h(T, act, li, i, w, k, rev, h, claimed, found);
}
// (rev, h: the window's orientation and key hash, computed by the caller)
__device__ inline u64 upsert_wave_h(const TableArgs& T, bool act, u32 li, u64 i, const u64* w, u32 k, bool rev, u64 h, bool& claimed, bool& found) {
    const int lane = threadIdx.x & 63;
    const u64 fp = (h >> 34) & T.fp_mask;
    const u64 myword_lo = ((u64)rev << 32) | (u64)(u32)i;
    auto eq = [&](u64 word) { return same_key_window(T.ks, word, w, rev); };
    // (1) to the first slot that is empty or carries the fingerprint
    u64 s = act ? home_slot(h, T.cap) : 0, word = EMPTY, probes = 0;
    claimed = false;
    bool cand = false;
    if (act) {
        for (; probes <= T.cap; ++probes) {
            u64 wv = load_relaxed(&T.tab[s].word);
            if (wv == EMPTY) {
                const u64 old = atomicCAS((unsigned long long*)&T.tab[s].word, (unsigned long long)EMPTY, (unsigned long long)((fp << 34) | myword_lo));
                if (old == EMPTY) { wave_agg_inc(T.n_distinct); claimed = true; break; }
                wv = old;
            }
            if ((wv >> 34) == fp) { word = wv; cand = true; break; }
            s = s + 1 == T.cap ? 0 : s + 1;
        }
        if (!claimed && !cand) { *T.probe_err = 1; s = ~0ull; }      // table full: cannot happen when it was sized from the true number of windows
    }
    // (2) heads and links
    const bool cross = rev != ((word & (1ull << 32)) != 0);
    const u32 rep = (u32)word;
    const u32 p_li = (u32)__shfl_up((int)li, 1, 64), p_rep = (u32)__shfl_up((int)rep, 1, 64);
    const int p_info = __shfl_up((int)((cand && !(word & (1ull << 33)) ? 1 : 0) | (cross ? 2 : 0)), 1, 64);
    const bool link = cand && !T.no_chain && lane > 0 && !(word & (1ull << 33)) && (p_info & 1) && p_li + 1 == li && ((p_info >> 1) & 1) == (int)cross &&
                      rep == (cross ? p_rep - 1u : p_rep + 1u);
    bool pass = false;
    if (cand) pass = link ? T.ks.mh[(u64)rep + (cross ? 0u : k - 1u)] == w[k - 1] : eq(word);
    const u64 heads = __ballot(!link), fails = __ballot(cand && !pass);
    found = pass;
    if (link && pass) {
        const int hp = 63 - __clzll((unsigned long long)(heads & ((2ull << lane) - 1ull)));      // the nearest head below: lane 0 is one
        found = ((fails >> hp) & ((2ull << (lane - hp)) - 1ull)) == 0;
    }
    if (T.link_ctr) { const u64 lm = __ballot(link && found); if (lane == 0 && lm) atomicAdd(T.link_ctr, (unsigned long long)__popcll(lm)); }
    if (cand && !found) {
        // another key behind this fingerprint, or a chain that broke in front of this lane: the plain walk, entered at this slot (a head has compared it already)
        const bool skip = !link;
        s = upsert_slot_from(T, h, myword_lo, eq, claimed, skip ? (s + 1 == T.cap ? 0 : s + 1) : s, probes + (skip ? 1 : 0));
        found = !claimed && s != ~0ull;
    }
    return s;
}

// Windows of the minimizers [i0, i1) of a batch -> counting table.  src/main.rs:756 — only reads with MORE than k minimizers contribute,
// so most minimizer indices start no window (15 kb reads at d = 0.002: 14 of 48), and with a partitioned table (replicated-sketch mode)
// most windows belong to other ranks.  One workgroup stages the OWN_SPAN + k - 1 hashes its span covers in LDS (coalesced), lists the
// local indices of the windows this rank has to insert, and then works the list off densely, so the long find-or-claim chains run on
// full wavefronts (0.80 -> 0.64 ms per 6.6 M windows against one thread per minimizer index).  Orientation, hash and the own side of the
// key comparison read the staged values.
#ifndef OWN_SPAN_V
#define OWN_SPAN_V 2048
#endif
constexpr int OWN_SPAN = OWN_SPAN_V;
__global__ __launch_bounds__(256) void insert_windows_kernel(TableArgs T, const u64* __restrict__ mh, const u32* __restrict__ mread,
                                                                   const u64* __restrict__ roff, u64 i0, u64 i1, u32 slot0, u64 first_ordinal,
                                                                   u32* __restrict__ cap_err, const u64* __restrict__ i1_dev, u64 n_lim) {
    extern __shared__ u64 sh_keys[];           // [OWN_SPAN + k] keys, then u16 list[OWN_SPAN], then the counter, then u8 cl[OWN_SPAN], then (partitioned table) 2 x u16 codes[OWN_SPAN + k]
    if (cap_err[1]) return;
    // i1_dev: launched behind the sketch of the same batch before the host knew how many minimizers it has (i1 = an upper bound the grid was
    // sized for): the count comes from the device, workgroups behind it have nothing to do
    if (i1_dev) { i1 = *i1_dev; if (i0 + (u64)blockIdx.x * OWN_SPAN >= i1) return; }
    const u32 k = T.ks.k;
    u16* const list = (u16*)(sh_keys + OWN_SPAN + k);
    u32* const n_own = (u32*)(list + OWN_SPAN);
    u8* const cl = (u8*)(n_own + 4);             // [OWN_SPAN] claim bytes of the span (T.claim)
    const u64 b0 = i0 + (u64)blockIdx.x * OWN_SPAN;
    const u64 lim = b0 + OWN_SPAN + k - 1 < i1 ? b0 + OWN_SPAN + k - 1 : i1;
    for (u64 t = b0 + threadIdx.x; t < lim; t += 256) sh_keys[t - b0] = mh[t];
    if (threadIdx.x == 0) *n_own = 0;
    if (T.claim) for (int u = threadIdx.x; u < OWN_SPAN / 8; u += 256) ((u64*)cl)[u] = 0;
    __syncthreads();
    // partitioned table: whose window is it?  From the codes of the staged hashes (see "owner codes"), or — no owner parameters — from the k values themselves
    const bool by_codes = T.own_world > 1 && T.own_thr != nullptr;
    const u16* mc = nullptr; u32 pc = 1; OwnerCodes oc{};
    if (by_codes) {
        u16* const ca = (u16*)(cl + OWN_SPAN);
        oc = owner_codes_of(OwnerSpec{T.own_world, T.own_thr});
        const u32 n_st = (u32)(lim - b0);
        mc = span_min_codes([&](u32 t) { return sh_keys[t]; }, [&](u32 t) { return t < n_st; }, OWN_SPAN + k - 1, k, oc, ca, ca + (OWN_SPAN + k), pc);
    }
#pragma unroll
    for (int u = 0; u < OWN_SPAN / 256; ++u) {
        const u32 li = u * 256 + threadIdx.x;
        const u64 i = b0 + li;
        bool mine = false;
        bool own = i + k <= i1;
        if (own && T.own_world > 1) {
            if (by_codes) { const u16 x = mc[li], y = mc[li + k - pc]; own = owner_of_code(x < y ? x : y, oc) == T.own_rank; }
            else own = window_owner(sh_keys + li, k, OwnerSpec{T.own_world, T.own_thr}) == T.own_rank;
        }
        if (own) {                                                        // ownership first: it needs no further loads
            const u32 slot = mread[i];
            const u64 rs = roff[slot], re = roff[slot + 1];
            mine = re - rs > k && i + k <= re;
        }
        const u64 m = __ballot(mine);
        u32 base = 0;
        if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(n_own, (u32)__popcll(m));
        base = __shfl(base, 0, 64);
        if (mine) list[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1))] = (u16)li;
    }
    __syncthreads();
    const u32 n = *n_own;
    if (threadIdx.x == 0 && n && T.own_world > 1) atomicAdd((unsigned long long*)ctr_shard(T.own_inserted), (unsigned long long)n);      // (checked against the senders' counts: partitioned tables only)
    for (u32 j0 = 0; j0 < n; j0 += 256) {          // (the same trip count for every lane: upsert_wave is a wave-wide call)
        const u32 j = j0 + threadIdx.x;
        bool act = j < n;
        const u32 li = act ? list[j] : 0u;
        const u64 i = b0 + li;
        u64 ord = 0;
        if (act) {
            const u32 slot = mread[i];
            const u64 win = i - roff[slot];
            if (win > WIN_MASK) { *cap_err = 1; act = false; }
            ord = ((first_ordinal + (slot - slot0)) << WIN_BITS) | win;
        }
        bool claimed, found;
        const u64 s = upsert_wave(T, act, li, i, sh_keys + li, k, claimed, found);
        if (claimed && T.claim) cl[li] = 1;
        if (found) {
            atomicAdd(&T.tab[s].count, 1u);
            push_ordinal(T, s, ord);
        }
    }
    if (T.claim) {                             // the span's claim bytes, 64 consecutive bytes per wave and store (a slice's last workgroup stops at its end: n_lim)
        __syncthreads();
        const u64 hi = i1 - b0 < (u64)OWN_SPAN ? i1 - b0 : (u64)OWN_SPAN;
        for (u32 li = threadIdx.x; li < hi && li < n_lim - (u64)blockIdx.x * OWN_SPAN; li += 256) T.claim[b0 + li] = cl[li];
    }
}

// ---- owner lists (replicated-sketch mode) -----------------------------------------------------------------------------------------
// The rank that sketched a batch also lists, per owning rank, the windows that rank owns (u32 index of the window's first minimizer,
// relative to the batch, and the index of its read in the batch: a pair of u32): 8 bytes per window shipped with the sketch, so that a
// receiver inserts exactly its windows instead of scanning every foreign sketch for them, and needs no minimizer -> read map of the
// foreign sketch either — the per-rank work no longer grows with the number of ranks.
// Two passes with per-block counts and a scan in between (deterministic bucket sizes, no same-address atomics on global counters).
constexpr int OWNL_SPAN = 2048;               // window starts per block: the span of hashes a receiving workgroup stages in LDS (16 KB + k values)
constexpr u32 OWNL_MAX_WORLD = 64;
struct OwnerBases { u64 b[OWNL_MAX_WORLD]; }; // start of every owner's bucket in the list
__device__ inline bool window_starts_at(const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i, u64 i1, u32 k) {
    if (i >= i1) return false;
    const u32 slot = mread[i];
    const u64 rs = roff[slot], re = roff[slot + 1];
    return re - rs > k && i + k <= re;
}
// The count pass also leaves every window start's owner in owner_of[] (0xFF: no window starts there) for the write pass.  The smallest hash
// of all the span's windows comes from LDS: the span's hashes are staged once and reduced by doubling (min over 2, 4, ... p <= k values; a
// window of k is two overlapping stretches of p) — read from HBM window by window it was 35 loads each, 1.4 ms per 6.6 M windows.
constexpr u32 OWNL_LDS_MAX_K = 1024;          // longer k: the plain loop (2 x (OWNL_SPAN + k) values have to fit the default 64 KB of dynamic LDS)
// window minima of a batch counted per bin of the value range (hist[OWNER_BINS], added to): what the measured owner table is made from
__global__ __launch_bounds__(256) void owner_bins_kernel(const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i0, u64 i1, u32 k, u64 mul,
                                                         unsigned long long* __restrict__ hist) {
    extern __shared__ u64 sh_min[];                   // k <= OWNL_LDS_MAX_K: two arrays of OWNL_SPAN + k - 1 codes (u16)
    const u64 b0 = i0 + (u64)blockIdx.x * OWNL_SPAN;
    const bool staged = k <= OWNL_LDS_MAX_K;
    OwnerCodes oc{}; oc.mul = mul;                    // (bins: the codes are the bins themselves)
    const u16* cur = nullptr; u32 p = 1;
    if (staged) { u16* const ca = (u16*)sh_min; cur = span_min_codes([&](u32 t) { return mh[b0 + t]; }, [&](u32 t) { return b0 + t < i1; }, OWNL_SPAN + k - 1, k, oc, ca, ca + (OWNL_SPAN + k), p); }
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int u = 0; u < OWNL_SPAN / 256; ++u) {
        const u32 li = u * 256 + threadIdx.x;
        const u64 i = b0 + li;
        u32 bin = 0xFFFFFFFFu;
        if (window_starts_at(mread, roff, i, i1, k)) {
            if (staged) { const u16 x = cur[li], y = cur[li + k - p]; bin = x < y ? x : y; }
            else { u64 m = mh[i]; for (u32 j = 1; j < k; ++j) { const u64 x = mh[i + j]; m = x < m ? x : m; } bin = owner_code(m, oc); }
        }
        // one atomic per distinct bin of the wave (the heavy bins are hit by several lanes of every wave)
        for (u64 todo = __ballot(bin != 0xFFFFFFFFu); todo;) {
            const u32 bb = (u32)__shfl((int)bin, __ffsll((unsigned long long)todo) - 1, 64);
            const u64 mm = __ballot(bin == bb);
            if (bin == bb && (mm & ((1ull << lane) - 1)) == 0) atomicAdd(&hist[bb], (unsigned long long)__popcll(mm));
            todo &= ~mm;
        }
    }
}
void launch_owner_bins(const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 k, u64 mul, u64* hist, hipStream_t s) {
    const size_t lds = k <= OWNL_LDS_MAX_K ? 2 * ((size_t)OWNL_SPAN + k) * sizeof(u16) : 0;
    if (i1 > i0) hipLaunchKernelGGL(owner_bins_kernel, dim3((unsigned)((i1 - i0 + OWNL_SPAN - 1) / OWNL_SPAN)), dim3(256), lds, s, mh, mread, roff, i0, i1, k, mul, (unsigned long long*)hist);
}
__global__ __launch_bounds__(256) void owner_list_count_kernel(const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i0, u64 i1,
                                                               u32 k, u32 world, const u64* thr, u32* __restrict__ blk_cnt, u8* __restrict__ owner_of) {
    extern __shared__ u64 sh_min[];               // owner parameters and k <= OWNL_LDS_MAX_K: two arrays of OWNL_SPAN + k - 1 owner codes (u16), see "owner codes"
    __shared__ u32 hist[OWNL_MAX_WORLD];
    if (threadIdx.x < world) hist[threadIdx.x] = 0;
    const u64 b0 = i0 + (u64)blockIdx.x * OWNL_SPAN;
    const OwnerSpec os{world, thr};
    const bool staged = k <= OWNL_LDS_MAX_K && thr != nullptr && world > 1;
    OwnerCodes oc{}; const u16* cur = nullptr; u32 p = 1;
    if (staged) {
        // the hashes are turned into codes as they are read: nothing but 2 x 2 bytes per element is staged (round 3 - 5: the u64 values, 33 KB of LDS per workgroup)
        oc = owner_codes_of(os);
        u16* const ca = (u16*)sh_min;
        cur = span_min_codes([&](u32 t) { return mh[b0 + t]; }, [&](u32 t) { return b0 + t < i1; }, OWNL_SPAN + k - 1, k, oc, ca, ca + (OWNL_SPAN + k), p);
    } else __syncthreads();
#pragma unroll
    for (int u = 0; u < OWNL_SPAN / 256; ++u) {
        const u32 li = u * 256 + threadIdx.x;
        const u64 i = b0 + li;
        u32 o = 0xFFu;
        if (window_starts_at(mread, roff, i, i1, k)) {
            if (staged) { const u16 x = cur[li], y = cur[li + k - p]; o = owner_of_code(x < y ? x : y, oc); }
            else o = window_owner(mh + i, k, os);
            atomicAdd(&hist[o], 1u);
        }
        if (i < i1) owner_of[i - i0] = (u8)o;
    }
    __syncthreads();
    if (threadIdx.x < world) blk_cnt[(size_t)blockIdx.x * world + threadIdx.x] = hist[threadIdx.x];
}
// Every owner's bucket comes out sorted by window start (the segments below are differences of neighbouring entries): the entries of a
// workgroup's span are ranked per owner in index order — lanes of a wave by ballots, the 32 (iteration, wave) groups by a prefix in LDS.
__global__ __launch_bounds__(256) void owner_list_write_kernel(const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i0, u64 i1,
                                                               u32 k, u32 world, const u64* thr, u32 slot0, const u64* __restrict__ blk_off, OwnerBases bases, u32* __restrict__ list,
                                                               const u8* __restrict__ owner_of, u32 direct_owner, u32* __restrict__ direct_dst) {
    // direct_owner (< world): that owner's bucket is not part of `list` (its bases entry is unused): it goes to direct_dst, the place the rank keeps its own
    // share of its own batch (api.inc, owner_lists_impl) — until round 5 the bucket was written to the list and copied there (368 MB per 19.5-Gbase batch at one rank) —,
    // or, direct_dst == null, NOWHERE: since round 6 the multi-GPU layer inserts a rank's own windows of its own batch with insert_windows_kernel, which finds them itself
    // (owner codes of the hashes it stages anyway), so that bucket is neither written nor read nor cut into spans
    constexpr int NG = OWNL_SPAN / 64;
    __shared__ u32 grp[NG][OWNL_MAX_WORLD];
    for (int t = threadIdx.x; t < NG * (int)OWNL_MAX_WORLD; t += 256) ((u32*)grp)[t] = 0;
    __syncthreads();
    const u64 b0 = i0 + (u64)blockIdx.x * OWNL_SPAN;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr u32 NONE = 0xFFFFFFFFu;
    u32 own[OWNL_SPAN / 256], rank[OWNL_SPAN / 256], slot_of[OWNL_SPAN / 256];
#pragma unroll
    for (int u = 0; u < OWNL_SPAN / 256; ++u) {
        const u64 i = b0 + u * 256 + threadIdx.x;
        u32 o = NONE, slot = 0;
        if (i < i1) { const u32 ob = owner_of[i - i0]; if (ob != 0xFFu) { o = ob; slot = mread[i]; } }      // (the count pass decided which starts are windows, and whose)
        u32 r = 0;
        for (u64 todo = __ballot(o != NONE); todo;) {
            const u32 oo = (u32)__shfl((int)o, __ffsll((unsigned long long)todo) - 1, 64);
            const u64 m = __ballot(o == oo);
            if (o == oo) { r = (u32)__popcll(m & ((1ull << lane) - 1)); if (r == 0) grp[u * 4 + wv][oo] = (u32)__popcll(m); }
            todo &= ~m;
        }
        own[u] = o; rank[u] = r; slot_of[u] = slot;
    }
    __syncthreads();
    if (threadIdx.x < world) { u32 run = 0; for (int g = 0; g < NG; ++g) { const u32 c = grp[g][threadIdx.x]; grp[g][threadIdx.x] = run; run += c; } }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < OWNL_SPAN / 256; ++u) {
        const u32 o = own[u];
        if (o == NONE || (o == direct_owner && !direct_dst)) continue;
        const u64 i = b0 + u * 256 + threadIdx.x;
        const u64 at = blk_off[(size_t)blockIdx.x * world + o] + grp[u * 4 + wv][o] + rank[u];
        uint2* const e = o == direct_owner ? (uint2*)direct_dst + at : (uint2*)list + (bases.b[o] + at);
        *e = make_uint2((u32)(i - i0), slot_of[u] - slot0);        // window start and its read, both relative to the batch
    }
}

// ---- segments: the hashes a rank's listed windows need, without the rest of the sketch -----------------------------------------------------------
// A bucket of the owner lists is sorted by window start w; the union of the windows' [w, w + k) is shipped as, per entry, the hashes it adds
// to the entries in front of it: all k when the window in front (same bucket) starts k or more earlier, else the last (w - w_prev) ones.  Both
// sides derive the same counts and their prefix from the list alone, so nothing but the list and the packed hashes travels.
// buckets: [n_buckets + 1] first entries (ascending); src / dst: hash index of window start 0 of every bucket.
struct SegBuckets { u64 start[OWNL_MAX_WORLD + 1]; u64 base[OWNL_MAX_WORLD]; u64 lim[OWNL_MAX_WORLD]; u32 n; u32 skip; };      // lim: hashes of the bucket's sketch (a window past
                                                                                                                                // it is skipped); skip: a bucket that ships nothing (the sender's own), or ~0
__device__ inline u32 seg_bucket_of(const SegBuckets& B, u64 j) {
    u32 lo = 0, hi = B.n - 1;
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (B.start[mid] <= j) lo = mid; else hi = mid - 1; }
    return lo;
}
// hashes entry j adds (b: its bucket)
__device__ inline u32 seg_add_of(const uint2* __restrict__ list, u64 j, u32 k, const SegBuckets& B, u32& b) {
    b = seg_bucket_of(B, j);
    if (b == B.skip) return 0;
    if (j && B.start[b] != j) { const u32 d = list[j].x - list[j - 1].x; if (d < k) return d; }
    return k;
}
// The counts are never stored: a pass over the list sums them per SEG_BLOCK entries (tile_scan_top_kernel turns the sums into bases), the copy pass derives them again
// and scans them inside its workgroup.  (Rounds 3 - 5 wrote a u32 count and a u64 prefix per entry and read both back: 1.98 ms per side for the 46 M entries of a
// 19.5-Gbase batch at eight ranks, 1.20 with the copy kernel below alone, profiles/r06_rank_w8.txt.)
constexpr u32 SEG_BLOCK = 4096;          // list entries per workgroup of the segment passes (one base per block: the single-workgroup scan of the bases stays short)
__global__ __launch_bounds__(256) void seg_sums_kernel(const uint2* __restrict__ list, u64 n, u32 k, SegBuckets B, u64* __restrict__ block_sum) {
    __shared__ u32 ws[4];
    u32 v = 0;
#pragma unroll
    for (int q = 0; q < SEG_BLOCK / 256; ++q) { const u64 j = (u64)blockIdx.x * SEG_BLOCK + q * 256 + threadIdx.x; u32 b; if (j < n) v += seg_add_of(list, j, k, B, b); }
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (u64)ws[0] + ws[1] + ws[2] + ws[3];
}
// out[b] = first payload index of bucket b, out[B.n] = total (one workgroup per value; block_base: the scanned sums, total: the scan's carry)
__global__ __launch_bounds__(256) void seg_pick_kernel(const uint2* __restrict__ list, u64 n, u32 k, SegBuckets B, const u64* __restrict__ block_base, const u64* __restrict__ total,
                                                       u64* __restrict__ out) {
    __shared__ u32 ws[4];
    const u32 bq = blockIdx.x;
    const u64 j1 = bq < B.n ? B.start[bq] : n;
    if (j1 >= n) { if (threadIdx.x == 0) out[bq] = total[0]; return; }
    const u64 j0 = j1 - j1 % SEG_BLOCK;
    u32 v = 0;
#pragma unroll
    for (int q = 0; q < SEG_BLOCK / 256; ++q) { const u64 j = j0 + q * 256 + threadIdx.x; u32 b; if (j < j1) v += seg_add_of(list, j, k, B, b); }
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[bq] = block_base[j0 / SEG_BLOCK] + ws[0] + ws[1] + ws[2] + ws[3];
}
// pack (to_store = 0): payload[prefix of j ..) <- the last (count of j) hashes of window j read from the store; scatter (to_store = 1): the other way.
// A workgroup takes SEG_BLOCK consecutive entries, 256 at a time: their parts of the payload are one contiguous stretch, which its threads walk element by element (the entry
// of an element: a search among the 256 prefixes in LDS) — every load and store of the payload side is coalesced, the store side runs along the windows' runs.  (Rounds
// 3 - 5: one thread per entry copying its values in a loop: a run's first window copies k values, the others one or two — every wave waited for its run heads, 8 bytes
// per lane and round trip.)
__global__ __launch_bounds__(256) void seg_copy_kernel(const uint2* __restrict__ list, u64 n, u32 k, SegBuckets B, const u64* __restrict__ block_base,
                                                       u64* __restrict__ store, u64* __restrict__ payload, u64 payload_n, u32 to_store) {
    __shared__ u32 lpre[256];
    __shared__ u64 laddr[256];
    __shared__ u32 tmp[8];
    u64 P0 = block_base[blockIdx.x];
#pragma unroll 1
    for (int q = 0; q < SEG_BLOCK / 256; ++q) {
        const u64 j = (u64)blockIdx.x * SEG_BLOCK + q * 256 + threadIdx.x;
        u32 a = 0, b = 0, x = 0;
        if (j < n) { a = seg_add_of(list, j, k, B, b); x = list[j].x; }
        u32 T;
        const u32 excl = block_excl_scan_256(a, tmp, T);          // (its barriers also keep this round's LDS writes behind the last round's reads)
        u64 at = ~0ull;
        if (j < n && !((u64)x + k > B.lim[b] || P0 + excl + a > payload_n)) at = B.base[b] + x + (k - a);      // (a wrong list: reported by the size check of the round / the count check of the insertion)
        lpre[threadIdx.x] = j < n ? excl : 0xFFFFFFFFu;
        laddr[threadIdx.x] = at;
        __syncthreads();
        for (u32 e = threadIdx.x; e < T; e += 256) {
            u32 lo = 0, hi = 255;                      // the last entry whose prefix is <= e (entries that add nothing share their prefix with the one behind them)
#pragma unroll
            for (int st = 0; st < 8; ++st) { const u32 mid = (lo + hi + 1) >> 1; if (lpre[mid] <= e) lo = mid; else hi = mid - 1; }
            const u64 src = laddr[lo];
            if (src == ~0ull) continue;
            u64* const h = store + src + (e - lpre[lo]);
            u64* const pp = payload + P0 + e;
            if (to_store) *h = *pp; else *pp = *h;
        }
        P0 += T;
    }
}
__device__ inline bool listed_check(u64 j) { return (((u32)j * 0x9E3779B1u) >> 28) == 0; }      // one list entry in 16
// inserts exactly the listed windows of the batch whose minimizers start at m0 (keys are read from the resident store)
// multi (non-null): the lists of SEVERAL batches in one launch — entry j belongs to batch b with multi[b].start <= j < multi[b + 1].start (n_multi batches
// and a closing entry); the per-batch arguments then come from the table.  (At 8 ranks and two chunks per step a rank inserts from 16 listed batches:
// 16 launches of ~0.4 M windows each.)
struct ListedBatch { u64 start, m0, m1, first_ordinal; const u32* list; u32 slot0, n_reads; };
// The per-entry kernel.  Its predecessor (rounds 3 - 5) waited on chains of dependent loads: four values per round trip of its hash loop (nine round trips at k = 35), k more
// for the owner check that one lane in 16 made (and every wave has such a lane): ~90 us per workgroup at full occupancy, 5.4 M windows/ms on a rank of eight
// (profiles/r06_rank_w8_ab.txt) against 10.8 M for the local kernel, whose values lie in LDS.  Here a window is read ONCE, sixteen values per round trip: hash and smallest
// value together (so EVERY entry's owner is re-derived, not one in 16), the batch of a wave's entries is found once per wave, and the walk is upsert_wave's: a bucket of the
// owner lists is sorted by window start and a rank's windows come in runs, so neighbouring lanes hold neighbouring windows, confirm each other as links, and their loads fall
// into the same cache lines.  8.4 M windows/ms.
__global__ __launch_bounds__(256) void insert_listed_entries_kernel(TableArgs T, const u64* __restrict__ mh, u32* __restrict__ mread, const u64* __restrict__ roff,
                                                                    u64 m0, u64 m1, const u32* __restrict__ list, u64 n, u32 slot0, u32 n_reads, u64 first_ordinal,
                                                                    u32* __restrict__ cap_err, const ListedBatch* __restrict__ multi, u32 n_multi) {
    if (cap_err[1]) return;
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = T.ks.k;
    bool ok = j < n;
    u64 i = 0, rs = 0; u32 slot = 0;
    u64 jl = j;
    if (multi) {
        // the batch of the wave's FIRST entry, searched once per wave on scalar loads; a lane behind the next batch's start (a wave across a boundary) searches for itself
        const u64 jw0 = (u64)blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
        u64 jw = ((u64)__builtin_amdgcn_readfirstlane((u32)(jw0 >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((u32)jw0);
        if (jw >= n) jw = n - 1;
        u32 lo = 0, hi = n_multi - 1;
        while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (multi[mid].start <= jw) lo = mid; else hi = mid - 1; }
        if (ok && multi[lo + 1].start <= j) {          // (multi[n_multi] is the closing entry: start = n)
            hi = n_multi - 1;
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (multi[mid].start <= j) lo = mid; else hi = mid - 1; }
        }
        const ListedBatch b = multi[lo];
        m0 = b.m0; m1 = b.m1; list = b.list; slot0 = b.slot0; n_reads = b.n_reads; first_ordinal = b.first_ordinal; jl = j - b.start;
    }
    if (ok) {
        const uint2 e = ((const uint2*)list)[jl];
        i = m0 + e.x; slot = slot0 + e.y; ok = e.y < n_reads && i + k <= m1;
    }
    const u64* const w = mh + i;
    bool rev = false; u64 h = 0;
    if (ok) {                                      // a wrong list is caught by the count check
        // one round trip: the read's offsets and the window's two ends (the orientation is decided by them unless they are equal)
        const u64 w_first = w[0], w_last = w[k - 1];
        rs = roff[slot]; const u64 re = roff[slot + 1];
        ok = i >= rs && re - rs > k && i + k <= re;
        if (ok) {
            rev = w_first != w_last ? w_first > w_last : window_reversed(w, k);
            u64 smallest;
            h = key_hash_window_hbm(w, k, rev, smallest);
            ok = owner_of_min(smallest, k, OwnerSpec{T.own_world, T.own_thr}) == T.own_rank;      // a sender that disagrees about the owner function: the count check fails
        }
    }
    wave_count_add(ok, T.own_inserted);
    const u64 win = i - rs;
    if (ok && win > WIN_MASK) { *cap_err = 1; ok = false; }
    const u64 ord = ((first_ordinal + (slot - slot0)) << WIN_BITS) | win;
    bool claimed, found;
    const u64 s = upsert_wave_h(T, ok, (u32)i, i, w, k, rev, h, claimed, found);      // (li = the store index: consecutive windows of a read are consecutive indices; no lane leaves early)
    if (claimed) { mread[i] = slot; if (T.claim) T.claim[i] = 1; }      // (the batch's bytes of the claim map were zeroed in front of the launch: api.inc, insert_resident_impl)
    if (found) {
        atomicAdd(&T.tab[s].count, 1u);
        push_ordinal(T, s, ord);
    }
}
// The lists are written span by span (owner_list_write_kernel: the entries of one OWNL_SPAN of window starts are contiguous, in any order
// inside it), so the receiver can work span-wise too: seg[q] = first list entry of span q or later (seg[] is pre-filled with n).
__global__ __launch_bounds__(256) void list_segments_kernel(const u32* __restrict__ list, u64 n, u32 n_spans, u32* __restrict__ seg) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 last = n_spans - 1;
    u32 b = list[2 * j] / OWNL_SPAN; if (b > last) b = last;                 // entries out of range are rejected by the insert kernel
    int pb = -1;
    if (j) { u32 q = list[2 * j - 2] / OWNL_SPAN; if (q > last) q = last; pb = (int)q; }
    for (int q = pb + 1; q <= (int)b; ++q) seg[q] = (u32)j;                 // all spans' loops together: n_spans stores
}
// inserts exactly the listed windows, one workgroup per span of window starts: the span's hashes are staged in LDS once (coalesced) and
// orientation, key hash and the own side of the key comparison read them from there — a rank's share of a foreign sketch is one window in
// `world`, read straight from HBM every one of them would fetch its k values over again
__global__ __launch_bounds__(256) void insert_listed_span_kernel(TableArgs T, const u64* __restrict__ mh, u32* __restrict__ mread, const u64* __restrict__ roff,
                                                                 u64 m0, u64 m1, const u32* __restrict__ list, const u32* __restrict__ seg, u64 n, u32 slot0,
                                                                 u32 n_reads, u64 first_ordinal, u32* __restrict__ cap_err) {
    extern __shared__ u64 sh_keys[];           // [OWNL_SPAN + k - 1] (+ one more, then u8 cl[OWNL_SPAN]: the span's claim bytes, T.claim)
    if (cap_err[1]) return;
    const u32 q = blockIdx.x, k = T.ks.k;
    const u32 s0 = seg[q], s1 = seg[q + 1];
    const u64 b0 = m0 + (u64)q * OWNL_SPAN;
    u8* const cl = (u8*)(sh_keys + OWNL_SPAN + k);
    const u64 span_n = m1 - b0 < (u64)OWNL_SPAN ? m1 - b0 : (u64)OWNL_SPAN;      // window starts of this span that exist
    if (s0 >= s1 || s1 > n) {                      // nothing listed here: no window of this span created a key (every byte of a batch's claim map is written by somebody)
        if (T.claim) for (u32 li = threadIdx.x; li < span_n; li += 256) T.claim[b0 + li] = 0;
        return;
    }
    const u64 lim = b0 + OWNL_SPAN + k - 1 < m1 ? b0 + OWNL_SPAN + k - 1 : m1;
    for (u64 t = b0 + threadIdx.x; t < lim; t += 256) sh_keys[t - b0] = mh[t];
    if (T.claim) for (int u = threadIdx.x; u < OWNL_SPAN / 8; u += 256) ((u64*)cl)[u] = 0;
    __syncthreads();
    for (u32 base = s0; base < s1; base += 256) {
        const u32 j = base + threadIdx.x;
        bool ok = j < s1;
        u32 li = 0; u64 i = 0;
        u32 slot = 0; u64 rs = 0;
        if (ok) {
            const uint2 e = ((const uint2*)list)[j];
            li = e.x; ok = li / OWNL_SPAN == q && e.y < n_reads; li -= q * OWNL_SPAN; i = b0 + li; slot = slot0 + e.y;
        }
        if (ok) {                                  // a wrong list is caught by the count check
            rs = roff[slot]; const u64 re = roff[slot + 1];
            ok = i >= rs && re - rs > k && i + k <= re && (!listed_check(j) || window_owner(sh_keys + li, k, OwnerSpec{T.own_world, T.own_thr}) == T.own_rank);
        }
        wave_count_add(ok, T.own_inserted);
        const u64 win = i - rs;
        if (ok && win > WIN_MASK) { *cap_err = 1; ok = false; }
        const u64 ord = ((first_ordinal + (slot - slot0)) << WIN_BITS) | win;
        bool claimed, found;
        const u64 s = upsert_wave(T, ok, li, i, sh_keys + li, k, claimed, found);      // (a wave-wide call: no lane leaves the loop early)
        if (claimed) { mread[i] = slot; if (T.claim) cl[li] = 1; }      // (mread: rep_ordinal() finds the representative's read through it; the rest of a listed batch's map is filled on demand)
        if (found) {
            atomicAdd(&T.tab[s].count, 1u);
            push_ordinal(T, s, ord);
        }
    }
    if (T.claim) {                                 // the span's claim bytes, 64 consecutive bytes per wave and store
        __syncthreads();
        for (u32 li = threadIdx.x; li < span_n; li += 256) T.claim[b0 + li] = cl[li];
    }
}
u32 owner_list_spans(u64 n_minimizers) { return (u32)((n_minimizers + OWNL_SPAN - 1) / OWNL_SPAN); }
// list: n pairs of u32; seg: owner_list_spans(m1 - m0) + 1 entries
void launch_list_segments(const u32* list, u64 n, u32 n_spans, u32* seg, hipStream_t s) {
    (void)hipMemsetD32Async((hipDeviceptr_t)seg, (int)(u32)n, (size_t)n_spans + 1, s);
    if (n && n_spans) hipLaunchKernelGGL(list_segments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, list, n, n_spans, seg);
}
void launch_owner_list_count(const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 k, u32 world, const u64* thr, u32* blk_cnt, u8* owner_of, hipStream_t s) {
    const size_t lds = k <= OWNL_LDS_MAX_K ? 2 * ((size_t)OWNL_SPAN + k) * sizeof(u16) : 0;
    if (i1 > i0) hipLaunchKernelGGL(owner_list_count_kernel, dim3((unsigned)((i1 - i0 + OWNL_SPAN - 1) / OWNL_SPAN)), dim3(256), lds, s, mh, mread, roff, i0, i1, k, world, thr, blk_cnt, owner_of);
}
void launch_owner_list_write(const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 k, u32 world, const u64* thr, u32 slot0, const u64* blk_off, const OwnerBases& bases, u32* list, const u8* owner_of, hipStream_t s,
                             u32 direct_owner = 0xFFFFFFFFu, u32* direct_dst = nullptr) {
    if (i1 > i0) hipLaunchKernelGGL(owner_list_write_kernel, dim3((unsigned)((i1 - i0 + OWNL_SPAN - 1) / OWNL_SPAN)), dim3(256), 0, s, mh, mread, roff, i0, i1, k, world, thr, slot0, blk_off, bases, list, owner_of, direct_owner, direct_dst);
}
// list: n pairs (window start, read), seg: launch_list_segments of it
// (95 registers = five waves per SIMD; forced to six or eight — 12 / 84 bytes of scratch — the kernel is slower: 5.87 / 7.00 against 5.48 ms per 46 M windows)
static void launch_listed_entries(const TableArgs& T, const u64* mh, u32* mread, const u64* roff, u64 m0, u64 m1, const u32* list, u64 n, u32 slot0, u32 n_reads, u64 first_ordinal,
                                  u32* cap_err, const ListedBatch* multi, u32 n_multi, hipStream_t s) {
    hipLaunchKernelGGL(insert_listed_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, T, mh, mread, roff, m0, m1, list, n, slot0, n_reads, first_ordinal, cap_err, multi, n_multi);
}
void launch_insert_listed(const TableArgs& T, const u64* mh, u32* mread, const u64* roff, u64 m0, u64 m1, const u32* list, const u32* seg, u64 n, u32 slot0,
                          u32 n_reads, u64 first_ordinal, u32* cap_err, hipStream_t s) {
    if (!n) return;
    const size_t lds = ((size_t)OWNL_SPAN + T.ks.k) * sizeof(u64) + OWNL_SPAN;      // hashes + the span's claim bytes
    // few listed windows per span (a rank's share of a sketch at 4+ ranks): staging every span of the sketch would mostly fetch hashes nobody needs, and
    // a workgroup would work off a few dozen entries; the per-entry kernel reads each window's values where they lie
    u64 per_span_min = 150;
    { const char* v = getenv("MDBG_LISTED_SPAN_MIN"); if (v) per_span_min = strtoull(v, nullptr, 10); }
    const bool sparse = n < (u64)owner_list_spans(m1 - m0) * per_span_min;
    if (seg && lds <= 64 * 1024 && !sparse)
        hipLaunchKernelGGL(insert_listed_span_kernel, dim3(owner_list_spans(m1 - m0)), dim3(256), lds, s, T, mh, mread, roff, m0, m1, list, seg, n, slot0, n_reads, first_ordinal, cap_err);
    else       // (also very long k: the span does not fit the default LDS window, every window reads its values from HBM)
        launch_listed_entries(T, mh, mread, roff, m0, m1, list, n, slot0, n_reads, first_ordinal, cap_err, nullptr, 0u, s);
}
// true: launch_insert_listed would take the per-entry kernel for this batch (then several such batches can share one launch, launch_insert_listed_multi)
bool listed_is_sparse(const TableArgs& T, u64 m0, u64 m1, u64 n) {
    u64 per_span_min = 150;
    { const char* v = getenv("MDBG_LISTED_SPAN_MIN"); if (v) per_span_min = strtoull(v, nullptr, 10); }
    return n < (u64)owner_list_spans(m1 - m0) * per_span_min || ((size_t)OWNL_SPAN + T.ks.k) * sizeof(u64) + OWNL_SPAN > 64 * 1024;
}
void launch_insert_listed_multi(const TableArgs& T, const u64* mh, u32* mread, const u64* roff, const ListedBatch* d_batches, u32 n_batches, u64 total, u32* cap_err, hipStream_t s) {
    if (!total) return;
    launch_listed_entries(T, mh, mread, roff, 0ull, 0ull, nullptr, total, 0u, 0u, 0ull, cap_err, d_batches, n_batches, s);
}

// Device-side twin of table_reserve(): flags the batch when the table is too small for it, so that the host can launch
// the insert speculatively and needs one round trip per batch instead of two.  Same rule as slots_for() in api.inc.
// The number of keys in the table is summed from its shards here (one launch less in front of every insertion).
// carry / over_max (null: not used): the insertion was launched behind the batch's own sketch, before the host looked at it — when the sketch has to
// be repeated (a slab or the store was too small, or the 2^32 limit) the insertion must not happen either.
__global__ __launch_bounds__(256) void reserve_check_kernel(const u64* __restrict__ distinct_shards, u64* __restrict__ n_distinct, const u64* __restrict__ batch_windows, u64 cap,
                                                            u32* __restrict__ too_small, const u64* __restrict__ carry, u64 store_cap, const u32* __restrict__ over_max) {
    __shared__ u64 ws[4];
    u64 v = 0;
    for (int i = threadIdx.x; i < CTR_SHARDS; i += 256) v += distinct_shards[i];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 nd = ws[0] + ws[1] + ws[2] + ws[3];
        *n_distinct = nd;
        const u64 n = nd + *batch_windows;
        bool skip = n + n / 2 + 1024 > cap;
        if (carry) skip = skip || *carry > store_cap || *carry >= 0xFFFFFFF0ull || (over_max && *over_max);
        *too_small = skip ? 1u : 0u;
    }
}

// count_windows_kernel + reserve_check_kernel in ONE launch (the insertion launched behind the batch's own sketch: two 8-microsecond kernels in a row in front of every
// insertion).  Every workgroup adds its reads' windows to *batch_windows and then to *done; the workgroup that brings *done to the grid size runs the check (the
// others' additions are visible to it: device-scope atomics, a fence on both sides) and sets *done back to zero for the next launch.
__global__ __launch_bounds__(1024) void count_reserve_kernel(const u64* __restrict__ roff, u32 slot0, u32 n_reads, u32 k, u64* __restrict__ batch_windows, u32* __restrict__ done,
                                                             const u64* __restrict__ distinct_shards, u64* __restrict__ n_distinct, u64 cap, u32* __restrict__ too_small,
                                                             const u64* __restrict__ carry, u64 store_cap, const u32* __restrict__ over_max) {
    __shared__ u64 ws[16];
    __shared__ u32 last;
    u64 w = 0;
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) { const u64 n = roff[slot0 + r + 1] - roff[slot0 + r]; if (n > k) w += n - k + 1; }
    for (int d = 32; d; d >>= 1) w += __shfl_down(w, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (int i = 0; i < 16; ++i) t += ws[i];
        if (t) atomicAdd((unsigned long long*)batch_windows, (unsigned long long)t);      // (at most 64 workgroups: two same-address atomics each, ~12 ns apiece)
        __threadfence();
        last = atomicAdd(done, 1u) + 1u == gridDim.x ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    u64 v = 0;
    for (int i = threadIdx.x; i < CTR_SHARDS; i += 1024) v += distinct_shards[i];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();                                   // (ws is read above by thread 0 only, long ago; the barrier is for its reuse)
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 nd = 0;
        for (int i = 0; i < 16; ++i) nd += ws[i];
        *n_distinct = nd;
        const u64 bw = __hip_atomic_load(batch_windows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 n = nd + bw;
        bool skip = n + n / 2 + 1024 > cap;                                                      // slots_for(), api.inc
        if (carry) skip = skip || *carry > store_cap || *carry >= 0xFFFFFFF0ull || (over_max && *over_max);
        *too_small = skip ? 1u : 0u;
        *done = 0;
    }
}
void launch_count_reserve(const u64* roff, u32 slot0, u32 n_reads, u32 k, u64* batch_windows, u32* done, const u64* distinct_shards, u64* n_distinct, u64 cap, u32* too_small,
                          const u64* carry, u64 store_cap, const u32* over_max, hipStream_t s) {
    hipLaunchKernelGGL(count_reserve_kernel, dim3(n_reads ? std::min<u32>(64u, (n_reads + 1023) / 1024) : 1), dim3(1024), 0, s, roff, slot0, n_reads, k, batch_windows, done, distinct_shards, n_distinct, cap,
                       too_small, carry, store_cap, over_max);
}

// routed records (k canonical u64, ordinal, key hash) sitting in the arena at record index r0..
__global__ __launch_bounds__(256) void insert_records_kernel(TableArgs T, u64 r0, u64 r1, u64* __restrict__ n_windows) {
    const u64 r = r0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r1) return;
    const u32 k = T.ks.k;
    const u64* key = T.ks.arena + r * (k + 2);
    const u64 h = key[k + 1];                  // computed by the sender (route_count_kernel)
    bool claimed;
    const u64 s = upsert_slot(T, h, (1ull << 33) | (u64)(u32)r, [&](u64 word) { return same_key_window(T.ks, word, key, false); }, claimed);   // record keys are canonical
    if (claimed || s == ~0ull) return;
    atomicAdd(&T.tab[s].count, 1u);
    push_ordinal(T, s, key[k]);
    (void)n_windows;
}

struct ZeroList { u64* p[6]; u64 n[6]; u64* set_p; u64 set_v; };
// z: small regions zeroed by the same launch (mdbg_reset: the shards of the key counter and three scalars — a launch of their own until round 6)
__global__ void clear_table_kernel(Slot* __restrict__ tab, u64 cap, u64* __restrict__ mx, u64 n_mx, ZeroList z) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    {
        const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
        for (int r = 0; r < 6; ++r) for (u64 i = i0; i < z.n[r]; i += stride) z.p[r][i] = 0;
        if (i0 == 0 && z.set_p) *z.set_p = z.set_v;
    }
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
        uint4* p = (uint4*)(tab + i);
        p[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);     // word, m1
        p[1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);                         // m2, count, pad
    }
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_mx; i += stride) mx[i] = EMPTY;
}

// grow: move every occupied slot of the old table into the new one (keys are unique: claim the first empty slot)
__global__ void rehash_kernel(const Slot* __restrict__ old, u64 old_cap, const u64* __restrict__ old_mx, TableArgs T) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= old_cap) return;
    const Slot e = old[i];
    if (e.word == EMPTY) return;
    const u64 h = key_hash_fn([&](u32 j) { return rep_elem(T.ks, e.word, j); }, T.ks.k);
    u64 s = home_slot(h, T.cap);
    for (;;) {
        const u64 oldw = atomicCAS((unsigned long long*)&T.tab[s].word, (unsigned long long)EMPTY, (unsigned long long)e.word);
        if (oldw == EMPTY) break;
        s = s + 1 == T.cap ? 0 : s + 1;
    }
    T.tab[s].m1 = e.m1; T.tab[s].m2 = e.m2; T.tab[s].count = e.count;
    for (u32 j = 0; j + 2 < T.A; ++j) T.mx[s * (T.A - 2) + j] = old_mx[i * (T.A - 2) + j];
}

// ---- finalize --------------------------------------------------------------------------------------
struct BatchTab {                 // batches sorted by first_ordinal (device copy)
    const u64* first_ordinal; const u32* n_reads; const u32* slot0; const u64* rank_base; u32 n;
    const u32* by_slot0; const u64* by_slot_first;   // the same batches sorted by slot0 (= call order): slot -> read ordinal
    const u64* by_m0; const u64* by_m0_rank;         // the batches sorted by position in the store: first minimizer index and its dense rank
    const u64* m0;                                   // in first_ordinal order: first minimizer index of the batch (dense index -> store index, claims_to_bits_kernel)
};
struct FinArgs {
    const Slot* tab; u64 cap; const u64* mx; u32 A; u32 casc; u32 k; u32 l;   // A: abundance filter; casc: ordinals tracked per slot (= A up to 8, else 1)
    const u64* mh; const u32* mpos; const u64* roff; const u32* mread; const u64* arena;
    BatchTab bt;
    u64* solid_list; u64* solid_count;       // compact list of solid slots (fin_mark -> fin_emit), any order
    u64* solid_dense;                        // dense ordered index of each listed slot's first sighting (-> its row, fin_order_kernel)
    const u64* n_solid_dev;                  // non-null: fin_order / fin_emit were launched for an ESTIMATED number of rows (their grid and the capacity of the
                                             // outputs) before the host knew the count: they take it from here and do nothing when it exceeds the estimate
    const u64* order;                        // non-null: slot of the node in row q (fin_emit then writes its rows in order: whole lines instead of
                                             // eleven scattered 2..8-byte stores per node)
    u64* o_row;                              // non-null: write node q at position q and its global row here (partitioned table)
    const u64* ath_override;                 // [Slot.pad - 1]: sighting whose metadata a node that wrapped its u16 abundance keeps (null: none)
    u64* bm_first; u64* bm_solid;            // bitmaps over dense ordered minimizer index
    u32 claims;                              // 1: by_first IS the insertion's claim map (TableArgs::claim) and dense index == store index: fin_mark only moves the marks of
                                             // keys whose first sighting is not their claimer (keys seen once — most — need nothing).  2 (round 6): the same map where the
                                             // dense order is NOT the store's (a partitioned table: the peers' regions lie between this rank's batches, batches out of
                                             // ordinal order): the map is indexed by STORE index, claims_to_bits_kernel turns it into the dense bitmaps
    u8* by_first; u8* by_solid;              // the same as one BYTE per index (zeroed): fin_mark sets bytes with plain stores — 3.9 M device-scope atomics on the
                                             // bitmaps were most of its time —, bytes_to_bits_kernel packs them into the bitmaps
    const u32* pre_first; const u32* pre_solid;   // exclusive popcount prefix per 64-bit word
    u64* sh_solid; u64* sh_wrapped; u64* sh_distinct;   // sharded counters (CTR_SHARDS u64 each)
    // outputs (device), node order = rank of first sighting among solid nodes
    u64* o_keys; u32* o_index; u16* o_abund; u32* o_seqlen; u16* o_shift; u64* o_shift_full;
    u64* o_src_read; u64* o_src_start; u64* o_src_end; u8* o_rev;
};

// ordinal -> (minimizer array index i, dense ordered index D)
__device__ inline void decode_ordinal(const FinArgs& F, u64 ord, u64& i, u64& D) {
    const u64 ro = ord >> WIN_BITS, win = ord & WIN_MASK;
    u32 lo = 0, hi = F.bt.n - 1;
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if (F.bt.first_ordinal[mid] <= ro) lo = mid; else hi = mid - 1; }
    const u32 s0 = F.bt.slot0[lo];
    const u32 slot = s0 + (u32)(ro - F.bt.first_ordinal[lo]);
    i = F.roff[slot] + win;
    D = F.bt.rank_base[lo] + (i - F.roff[s0]);
}
// Dense ordered index of the minimizer at store index i (the window starting there): batches keep their order inside the store, so
// this needs neither the read map nor the read offsets.  Dense indices are ordered like the ordinals they stand for.
__device__ inline u64 dense_of_index(const FinArgs& F, u64 i) {
    u32 lo = 0, hi = F.bt.n - 1;
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if (F.bt.by_m0[mid] <= i) lo = mid; else hi = mid - 1; }
    return F.bt.by_m0_rank[lo] + (i - F.bt.by_m0[lo]);
}
// ordinal of the occurrence that claimed the slot (it did no count / ordinal atomics)
__device__ inline u64 rep_ordinal(const FinArgs& F, u64 word) {
    const u32 rep = (u32)word;
    if (word & (1ull << 33)) return F.arena[(u64)rep * (F.k + 2) + F.k];
    const u32 slot = F.mread[rep];
    u32 lo = 0, hi = F.bt.n - 1;
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if (F.bt.by_slot0[mid] <= slot) lo = mid; else hi = mid - 1; }
    return ((F.bt.by_slot_first[lo] + (slot - F.bt.by_slot0[lo])) << WIN_BITS) | ((u64)rep - F.roff[slot]);
}
struct SlotView { u32 count; u64 first, ath; bool solid; };
// merges the claimer back in: total count, smallest ordinal, A-th smallest ordinal (valid when count >= A)
// casc: number of smallest ordinals the table tracked (= A for A <= MDBG_CASCADE_MAX; 1 for larger A, whose A-th sighting comes from
// the re-scan of resolve_wrapped through ath_override)
__device__ inline SlotView slot_view(const Slot& e, u64 s, const u64* mx, u32 casc, u32 A_filter, u64 r, const u64* ath_override = nullptr) {
    SlotView v;
    v.count = e.count + 1u;
    v.first = r < e.m1 ? r : e.m1;
    const u32 A = casc;
    if (A == 1) v.ath = v.first;
    else {
        const u64 prev = A == 2 ? e.m1 : A == 3 ? e.m2 : mx[s * (A - 2) + (A - 4)];      // (A-1)-th smallest of the others
        const u64 last = A == 2 ? e.m2 : mx[s * (A - 2) + (A - 3)];                      // A-th smallest of the others
        v.ath = r < prev ? prev : (r < last ? r : last);
    }
    v.solid = A_filter == 1 || (u16)v.count >= (u16)A_filter;                             // src/main.rs:922-929 (u16 abundance)
    if (e.pad && ath_override) v.ath = ath_override[e.pad - 1];                            // see wrap_list_kernel
    return v;
}

// FIN_SPT slots per thread, all requested before the first is looked at: the kernel is a chain of dependent round trips (slot -> read
// offsets of the smallest ordinal -> bitmap atomic), and with one slot per thread its 9,700 workgroups went through it 19 deep
constexpr int FIN_SPT = 4;
__global__ __launch_bounds__(1024) void fin_mark_kernel(FinArgs F) {
    __shared__ u32 wcnt[16 * FIN_SPT];
    __shared__ u64 bbase;
    const u64 s0 = (u64)blockIdx.x * (1024 * FIN_SPT) + threadIdx.x;
    Slot e[FIN_SPT];
#pragma unroll
    for (int u = 0; u < FIN_SPT; ++u) { const u64 s = s0 + 1024ull * u; e[u].word = EMPTY; if (s < F.cap) e[u] = F.tab[s]; }
    u32 n_occ = 0, n_wrapped = 0; bool solid[FIN_SPT]; u64 m[FIN_SPT], dense[FIN_SPT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < FIN_SPT; ++u) {
        solid[u] = false;
        u64 D = 0;
        if (e[u].word != EMPTY) {
            const u32 count = e[u].count + 1u;
            ++n_occ; solid[u] = F.A == 1 || (u16)count >= (u16)F.A; n_wrapped += count >= 65536u ? 1u : 0u;          // as slot_view
            // first sighting: the claimer's window or the smallest ordinal the others pushed.  Most keys are seen once (sequencing
            // errors), and the claimer's dense index follows from `rep` alone: no read map / offset lookups for them
            if (F.claims) {
                // the claimer's byte is set already (insert_windows_kernel); a key seen again may have an earlier sighting: move the mark there.  201 M scattered byte
                // stores into a 721 MB map were 10 ms of the human table's finalize (1.3 TB/s); 183 M of those keys are seen once and cost nothing here
                // ONE map in this mode: bit 0 = first sighting, bit 1 = the key is solid (no second map to zero: 2.2 ms per finalize of the human table); with the batches in ordinal order — the mode's
                // condition — a first sighting never moves forward
                // (F.claims == 2: the map is indexed by STORE index, the dense index of the claimer comes from the batch table)
                const u64 ic = (u32)e[u].word;
                const u64 Dc = F.claims == 2 ? dense_of_index(F, ic) : ic;
                u64 at = ic;
                D = Dc;
                if (e[u].count) { u64 i, D1; decode_ordinal(F, e[u].m1, i, D1); if (D1 < Dc) { D = D1; at = i; F.by_first[ic] = 0; } }
                // (a key seen again always rewrites its byte: its solid bit may have to GO — a u16 abundance that wrapped below minabund between two finalize calls, round-5 advice)
                if (e[u].count || solid[u]) F.by_first[at] = solid[u] ? 3 : 1;
            } else {
            if (e[u].word & (1ull << 33)) { u64 i; const u64 ro = rep_ordinal(F, e[u].word); decode_ordinal(F, ro < e[u].m1 ? ro : e[u].m1, i, D); }   // routed record
            else {
                D = dense_of_index(F, (u32)e[u].word);
                if (e[u].count) { u64 i, D1; decode_ordinal(F, e[u].m1, i, D1); if (D1 < D) D = D1; }
            }
            F.by_first[D] = 1;                       // (distinct keys have distinct first sightings: nobody else writes this byte)
            if (solid[u]) F.by_solid[D] = 1;
            }
        }
        dense[u] = D;
        m[u] = __ballot(solid[u]);
        if (lane == 0) wcnt[16 * u + wv] = (u32)__popcll(m[u]);
    }
    for (int d = 32; d; d >>= 1) { n_occ += __shfl_down(n_occ, d, 64); n_wrapped += __shfl_down(n_wrapped, d, 64); }
    if (lane == 0) {
        if (n_occ) atomicAdd((unsigned long long*)ctr_shard(F.sh_distinct), (unsigned long long)n_occ);
        if (n_wrapped) atomicAdd((unsigned long long*)ctr_shard(F.sh_wrapped), (unsigned long long)n_wrapped);
    }
    // compact list of solid slots (any order): one allocation atomic per block
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0;
        for (int i = 0; i < 16 * FIN_SPT; ++i) { const u32 c = wcnt[i]; wcnt[i] = tot; tot += c; }
        bbase = tot ? atomicAdd((unsigned long long*)F.solid_count, (unsigned long long)tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < FIN_SPT; ++u)
        if (solid[u]) {
            const u64 j = bbase + wcnt[16 * u + wv] + __popcll(m[u] & ((1ull << lane) - 1));
            F.solid_list[j] = s0 + 1024ull * u; F.solid_dense[j] = dense[u];
        }
}
// fin_mark_kernel for the claim-map mode (F.claims), in two passes over the workgroup's 4,096 slots.  A slot needs more than a look only when its key was seen again
// (the first sighting may lie in front of the claimer: one scattered read, up to two scattered byte stores) or is solid (a row of the node table): 9 % of the human
// table's slots, which in fin_mark_kernel sit spread over every wave, so every wave waits for its few slow lanes on each of its four rounds.  Here the first pass
// only looks (word and count) and lists those slots in LDS; the second pass works the list off on full waves, and lists the solid ones once more for the ONE
// allocation atomic per workgroup (the counter is a single address: ~12 ns per atomic, serialised).
__global__ __launch_bounds__(1024) void fin_mark_claims_kernel(FinArgs F) {
    constexpr u32 SPAN = 1024 * FIN_SPT;
    __shared__ u16 lst[SPAN], sol_li[SPAN];
    __shared__ u32 sol_D[SPAN];
    __shared__ u32 n_lst, n_sol;
    __shared__ u64 bbase;
    const u64 b0 = (u64)blockIdx.x * SPAN;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { n_lst = 0; n_sol = 0; }
    __syncthreads();
    u32 n_occ = 0, n_wrapped = 0;
#pragma unroll
    for (int u = 0; u < FIN_SPT; ++u) {
        const u32 li = (u32)u * 1024u + threadIdx.x;
        const u64 s = b0 + li;
        bool slow = false;
        if (s < F.cap) {
            const u64 word = F.tab[s].word;
            if (word != EMPTY) {
                const u32 c0 = F.tab[s].count, count = c0 + 1u;
                ++n_occ; n_wrapped += count >= 65536u ? 1u : 0u;
                slow = c0 != 0 || F.A == 1 || (u16)count >= (u16)F.A;          // seen again, or solid (as slot_view)
            }
        }
        const u64 mk = __ballot(slow);
        u32 base = 0;
        if (lane == 0 && mk) base = atomicAdd(&n_lst, (u32)__popcll(mk));
        base = (u32)__shfl((int)base, 0, 64);
        if (slow) lst[base + (u32)__popcll(mk & ((1ull << lane) - 1ull))] = (u16)li;
    }
    for (int d = 32; d; d >>= 1) { n_occ += __shfl_down(n_occ, d, 64); n_wrapped += __shfl_down(n_wrapped, d, 64); }
    if (lane == 0) {
        if (n_occ) atomicAdd((unsigned long long*)ctr_shard(F.sh_distinct), (unsigned long long)n_occ);
        if (n_wrapped) atomicAdd((unsigned long long*)ctr_shard(F.sh_wrapped), (unsigned long long)n_wrapped);
    }
    __syncthreads();
    const u32 n = n_lst;
    for (u32 t0 = 0; t0 < n; t0 += 1024) {            // (the same trip count for every lane: ballots inside)
        const u32 t = t0 + threadIdx.x;
        bool solid = false; u32 li = 0; u64 D = 0;
        if (t < n) {
            li = lst[t];
            const Slot e = F.tab[b0 + li];
            const u32 count = e.count + 1u;
            solid = F.A == 1 || (u16)count >= (u16)F.A;
            // the claimer's byte is set already (insert_windows_kernel); a key seen again may have an earlier sighting: move the mark there.  ONE map in this mode: bit 0 =
            // first sighting, bit 1 = the key is solid (see fin_mark_kernel)
            const u64 ic = (u32)e.word;                                    // store index of the claimer; F.claims == 2: the map is indexed by store index, not by dense index
            const u64 Dc = F.claims == 2 ? dense_of_index(F, ic) : ic;
            u64 at = ic;
            D = Dc;
            if (e.count) { u64 i, D1; decode_ordinal(F, e.m1, i, D1); if (D1 < Dc) { D = D1; at = i; F.by_first[ic] = 0; } }
            // every listed slot (seen again, or solid) rewrites its byte: the solid bit is also CLEARED — solidity is (u16)count >= (u16)A, which turns false again when the
            // abundance wraps (count 65535 finalized as solid, more batches, count 65536 = u16 0, finalized again: round-5 advice; the byte-map path zeroes its maps per finalize)
            F.by_first[at] = solid ? 3 : 1;
        }
        const u64 mk = __ballot(solid);
        u32 base = 0;
        if (lane == 0 && mk) base = atomicAdd(&n_sol, (u32)__popcll(mk));
        base = (u32)__shfl((int)base, 0, 64);
        if (solid) { const u32 q = base + (u32)__popcll(mk & ((1ull << lane) - 1ull)); sol_li[q] = (u16)li; sol_D[q] = (u32)D; }
    }
    __syncthreads();
    const u32 ns = n_sol;
    if (threadIdx.x == 0) bbase = ns ? atomicAdd((unsigned long long*)F.solid_count, (unsigned long long)ns) : 0;
    __syncthreads();
    for (u32 t = threadIdx.x; t < ns; t += 1024) { F.solid_list[bbase + t] = b0 + sol_li[t]; F.solid_dense[bbase + t] = (u64)sol_D[t]; }
}
// bitmap word w <- bit i = (byte 64 w + i != 0), for both maps; one thread per word (four 16-byte loads per map).  by1 == null: ONE map whose bytes hold bit 0 = first
// sighting, bit 1 = solid (the claim-map mode of fin_mark_kernel).  Bits at or behind n_bits are cleared (the claim map's bytes behind the store's end are whatever
// the allocation holds: masked here instead of being zeroed by a fill in front of every finalize).  block_sum (non-null): the popcounts of the workgroup's 1,024 words of
// either bitmap — what popc_block_kernel would compute in a launch of its own: [blockIdx] and [n_blocks + blockIdx].
__global__ __launch_bounds__(1024) void bytes_to_bits_kernel(const u8* __restrict__ by0, const u8* __restrict__ by1, u64 n_words, u64 n_bits, u64* __restrict__ bm0, u64* __restrict__ bm1,
                                                             u32* __restrict__ block_sum, u32 n_blocks) {
    __shared__ u32 ws[2][16];
    const u64 w = (u64)blockIdx.x * 1024 + threadIdx.x;
    auto pack = [](const u8* p, u32 shift) -> u64 {
        u64 out = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = ((const uint4*)p)[q];
            const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) out |= (u64)(((((x[j] >> shift) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (16 * q + 4 * j);      // four bytes -> four bits, byte 0 lowest
        }
        return out;
    };
    u64 a = 0, b = 0;
    if (w < n_words) {
        a = pack(by0 + 64 * w, 0); b = by1 ? pack(by1 + 64 * w, 0) : pack(by0 + 64 * w, 1);
        if (64 * w + 64 > n_bits) { const u64 keep = 64 * w >= n_bits ? 0ull : (1ull << (n_bits - 64 * w)) - 1ull; a &= keep; b &= keep; }
        bm0[w] = a; bm1[w] = b;
    }
    if (!block_sum) return;
    u32 v0 = (u32)__popcll(a), v1 = (u32)__popcll(b);
    for (int d = 32; d; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); }
    if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = v0; ws[1][threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x < 2) { u32 t = 0; for (int i = 0; i < 16; ++i) t += ws[threadIdx.x][i]; block_sum[threadIdx.x * n_blocks + blockIdx.x] = t; }
}
// block_sum: null, or 2 * ceil(n_words / 1024) u32 (then launch_popc_prefix2 may skip its first kernel: have_block_sums)
void launch_bytes_to_bits(const u8* by0, const u8* by1, u64 n_words, u64 n_bits, u64* bm0, u64* bm1, u32* block_sum, hipStream_t s) {
    const u32 nb = (u32)((n_words + 1023) / 1024);
    if (n_words) hipLaunchKernelGGL(bytes_to_bits_kernel, dim3(nb), dim3(1024), 0, s, by0, by1, n_words, n_bits, bm0, bm1, block_sum, nb);
}
// F.claims == 2: the claim map is indexed by STORE index, the bitmaps by dense ordered index (the batches in first-ordinal order): one wave per 64 dense indices — lane l reads
// its bytes at its batch's place in the store (consecutive lanes read consecutive bytes except across a batch boundary).  ~1 byte read per resident minimizer; the per-block popcounts are left to popc_block_kernel (a partitioned table merges the bitmaps over the ranks first).
__global__ __launch_bounds__(256) void claims_to_bits_kernel(FinArgs F, u64 n_words, u64 n_bits, u64* __restrict__ bm0, u64* __restrict__ bm1) {
    // a lane takes EIGHT dense indices (one byte of either bitmap): one 8-byte load where the eight lie in one batch (all but the groups across a batch boundary), its eight
    // bits 0 and eight bits 1 gathered by a multiplication, stored as a byte: a wave reads 512 bytes and writes 64 + 64 per round.  (Until round 6 a lane read one byte and
    // two ballots made the words: 64 bytes per wave and load, 1.2 ms of a rank-of-eight's finalize, which runs over the WHOLE index space.)
    u32 bi = 0; u64 lo = 0, hi = 0, m0 = 0;                                    // the batch [lo, hi) of dense indices the lane looked at last
    auto locate = [&](u64 D) {
        if (D >= lo && D < hi) return;
        u32 a = 0, z = F.bt.n - 1;
        while (a < z) { const u32 mid = a + ((z - a + 1) >> 1); if (F.bt.rank_base[mid] <= D) a = mid; else z = mid - 1; }
        bi = a; lo = F.bt.rank_base[bi]; hi = bi + 1 < F.bt.n ? F.bt.rank_base[bi + 1] : n_bits; m0 = F.bt.m0[bi];
        // (batches without a minimizer share their rank base with the batch behind them: the search ends on the last of them, whose range [lo, hi) holds D)
    };
    const u64 n_groups = n_words * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u64 g = ((u64)blockIdx.x * 4 + u) * 256 + threadIdx.x;          // (consecutive lanes: consecutive groups)
        if (g >= n_groups) break;
        const u64 D = 8 * g;
        u64 x = 0;
        if (D < n_bits) {
            locate(D);
            if (D + 8 <= hi) {                          // eight bytes at any alignment: the two aligned words around them (the second one is the next lane's first: a hit)
                const u8* const p = F.by_first + m0 + (D - lo);
                const u32 o = (u32)((uintptr_t)p & 7u) * 8u;
                const u64* const q = (const u64*)(p - (o >> 3));
                x = q[0];
                if (o) x = (x >> o) | (q[1] << (64u - o));
            }
            else for (u32 t = 0; t < 8 && D + t < n_bits; ++t) { locate(D + t); x |= (u64)F.by_first[m0 + (D + t - lo)] << (8 * t); }
        }
        ((u8*)bm0)[g] = (u8)(((x & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
        ((u8*)bm1)[g] = (u8)((((x >> 1) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
    }
}
void launch_claims_to_bits(const FinArgs& F, u64 n_words, u64 n_bits, u64* bm0, u64* bm1, hipStream_t s) {
    if (n_words) hipLaunchKernelGGL(claims_to_bits_kernel, dim3((unsigned)((n_words * 8 + 1023) / 1024)), dim3(256), 0, s, F, n_words, n_bits, bm0, bm1);
}
// row of every listed solid slot (rank of its first sighting among the solid ones) -> order[row] = slot
__global__ __launch_bounds__(256) void fin_order_kernel(FinArgs F, u64 n_solid, u64* __restrict__ order) {
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (F.n_solid_dev) { const u64 n = *F.n_solid_dev; if (n > n_solid) return; n_solid = n; }      // (n_solid: the estimate the launch was sized for)
    if (q >= n_solid) return;
    const u64 D = F.solid_dense[q];
    order[F.pre_solid[D >> 6] + __popcll(F.bm_solid[D >> 6] & ((1ull << (D & 63)) - 1))] = F.solid_list[q];
}

// ---- nodes whose u16 abundance wrapped (src/main.rs:676-684) ------------------------------------------
// The reference refreshes seqlen / shift (and writes the .sequences line) whenever the abundance BEFORE the increment
// equals minabund - 1.  The abundance is a u16 that wraps in release builds, so for a k-min-mer seen c >= 65536 + A times
// the entry ends up describing sighting j* = A + 65536 * floor((c - A) / 65536), not the A-th.  The min-cascade of the
// table only knows the A smallest ordinals; for these (rare, extremely repetitive) keys the exact j*-th smallest ordinal is
// recovered here: list them (Slot.pad = rank + 1), re-scan the resident windows (or routed records) collecting the
// ordinals of exactly those keys, sort each list, pick element j* - 1.
// all_solid: minabund exceeds what the slots track (MDBG_CASCADE_MAX): EVERY solid node gets its j*-th sighting this way.
__global__ __launch_bounds__(256) void wrap_list_kernel(Slot* __restrict__ tab, u64 cap, u32 A, bool all_solid, u64* __restrict__ w_jstar, u32* __restrict__ w_count,
                                                        unsigned long long* __restrict__ counters /* [0] nodes, [1] occurrences */) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap) return;
    const u64 word = tab[s].word;
    if (word == EMPTY) return;
    const u32 count = tab[s].count + 1u;
    if (count < A || (!all_solid && count - A < 65536u) || !(A == 1 || (u16)count >= (u16)A)) {
        // not listed (any more): a rank left by an earlier finalize (finalize -> ingest -> finalize without reset, the node's u16 abundance
        // has wrapped below minabund meanwhile) would send the scan kernels to another node's segment or past the lists
        if (tab[s].pad) tab[s].pad = 0;
        return;
    }
    const u32 r = (u32)atomicAdd(&counters[0], 1ull);
    atomicAdd(&counters[1], (unsigned long long)count);
    w_count[r] = count; w_jstar[r] = (u64)A + 65536ull * ((count - A) / 65536u);
    tab[s].pad = r + 1;
}
// lookup without insertion: slot of a key that is known to be in the table, or ~0 (keys of other ranks)
template <class EqFn>
__device__ inline u64 find_slot(const TableArgs& T, u64 h, EqFn same_key) {
    const u64 fp = (h >> 34) & T.fp_mask;
    u64 s = home_slot(h, T.cap);
    for (;;) {
        const u64 w = load_relaxed(&T.tab[s].word);
        if (w == EMPTY) return ~0ull;
        if ((w >> 34) == fp && same_key(w)) return s;
        s = s + 1 == T.cap ? 0 : s + 1;
    }
}
__global__ __launch_bounds__(256) void wrap_scan_windows_kernel(TableArgs T, const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff,
                                                                u64 i0, u64 i1, u32 slot0, u64 first_ordinal, const u32* __restrict__ w_start,
                                                                u32* __restrict__ w_fill, u64* __restrict__ occ) {
    extern __shared__ u64 sh_keys[];
    const u32 k = T.ks.k;
    const u64 b0 = i0 + (u64)blockIdx.x * 256;
    const u64 lim = b0 + 256 + k - 1 < i1 ? b0 + 256 + k - 1 : i1;
    for (u64 t = b0 + threadIdx.x; t < lim; t += 256) sh_keys[t - b0] = mh[t];
    const u64 i = b0 + threadIdx.x;
    bool active = i < i1;
    u64 ord = 0;
    if (active) {
        const u32 slot = mread[i];
        const u64 rs = roff[slot], re = roff[slot + 1];
        active = re - rs > k && i + k <= re && i - rs <= WIN_MASK;
        ord = ((first_ordinal + (slot - slot0)) << WIN_BITS) | (i - rs);
    }
    __syncthreads();
    if (!active) return;
    const u64* w = sh_keys + threadIdx.x;
    if (T.own_world > 1 && window_owner(w, k, OwnerSpec{T.own_world, T.own_thr}) != T.own_rank) return;
    const bool rev = window_reversed(w, k);
    const u64 s = find_slot(T, key_hash_window(w, k, rev), [&](u64 word) { return same_key_window(T.ks, word, w, rev); });
    if (s == ~0ull) return;
    const u32 pad = T.tab[s].pad;
    if (pad) occ[w_start[pad - 1] + atomicAdd(&w_fill[pad - 1], 1u)] = ord;
}
// the same over the LISTED windows of a batch (a foreign sketch of which only the listed windows' hashes are resident)
__global__ __launch_bounds__(256) void wrap_scan_listed_kernel(TableArgs T, const u64* __restrict__ mh, const u64* __restrict__ roff, u64 m0, u64 m1, const u32* __restrict__ list, u64 n,
                                                               u32 slot0, u32 n_reads, u64 first_ordinal, const u32* __restrict__ w_start, u32* __restrict__ w_fill, u64* __restrict__ occ) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 k = T.ks.k;
    const uint2 e = ((const uint2*)list)[j];
    const u64 i = m0 + e.x; const u32 slot = slot0 + e.y;
    if (e.y >= n_reads || i + k > m1) return;
    const u64 rs = roff[slot], re = roff[slot + 1];
    if (!(i >= rs && re - rs > k && i + k <= re && i - rs <= WIN_MASK)) return;
    const u64* w = mh + i;
    if (window_owner(w, k, OwnerSpec{T.own_world, T.own_thr}) != T.own_rank) return;
    const bool rev = window_reversed(w, k);
    const u64 s = find_slot(T, key_hash_window(w, k, rev), [&](u64 word) { return same_key_window(T.ks, word, w, rev); });
    if (s == ~0ull) return;
    const u32 pad = T.tab[s].pad;
    if (pad) occ[w_start[pad - 1] + atomicAdd(&w_fill[pad - 1], 1u)] = ((first_ordinal + e.y) << WIN_BITS) | (i - rs);
}
// --read_stats (src/main.rs:939-1004): abundance of the k-min-mer that starts at minimizer i in the FILTERED table, 0 when it
// is absent or below the abundance filter; NO_WINDOW where no window starts (fewer than k minimizers left in the read, or
// a read with at most k minimizers: src/main.rs:950, strictly more than k).
constexpr u32 NO_WINDOW = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void query_windows_kernel(TableArgs T, u32 A_filter, const u64* __restrict__ mh, const u32* __restrict__ mread,
                                                            const u64* __restrict__ roff, u64 i0, u64 i1, u32* __restrict__ out) {
    extern __shared__ u64 sh_keys[];
    const u32 k = T.ks.k;
    const u64 b0 = i0 + (u64)blockIdx.x * 256;
    const u64 lim = b0 + 256 + k - 1 < i1 ? b0 + 256 + k - 1 : i1;
    for (u64 t = b0 + threadIdx.x; t < lim; t += 256) sh_keys[t - b0] = mh[t];
    const u64 i = b0 + threadIdx.x;
    bool active = i < i1;
    if (active) {
        const u32 slot = mread[i];
        const u64 rs = roff[slot], re = roff[slot + 1];
        active = re - rs > k && i + k <= re;
        if (!active) out[i - i0] = NO_WINDOW;
    }
    __syncthreads();
    if (!active) return;
    const u64* w = sh_keys + threadIdx.x;
    const bool rev = window_reversed(w, k);
    u32 ab = 0;
    if (T.cap) {
        const u64 s = find_slot(T, key_hash_window(w, k, rev), [&](u64 word) { return same_key_window(T.ks, word, w, rev); });
        if (s != ~0ull) {
            const u32 count = T.tab[s].count + 1u;
            if (A_filter == 1 || (u16)count >= (u16)A_filter) ab = (u16)count;      // dbg_nodes.retain (main.rs:927), u16 abundance
        }
    }
    out[i - i0] = ab;
}
void launch_query_windows(const TableArgs& T, u32 A_filter, const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32* out, hipStream_t s) {
    if (i1 <= i0) return;
    const u64 n = i1 - i0;
    hipLaunchKernelGGL(query_windows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (256 + T.ks.k) * 8, s, T, A_filter, mh, mread, roff, i0, i1, out);
}
__global__ __launch_bounds__(256) void wrap_scan_records_kernel(TableArgs T, u64 n_records, const u32* __restrict__ w_start, u32* __restrict__ w_fill, u64* __restrict__ occ) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const u32 k = T.ks.k;
    const u64* key = T.ks.arena + r * (k + 2);
    const u64 s = find_slot(T, key[k + 1], [&](u64 word) { return same_key_window(T.ks, word, key, false); });
    if (s == ~0ull) return;
    const u32 pad = T.tab[s].pad;
    if (pad) occ[w_start[pad - 1] + atomicAdd(&w_fill[pad - 1], 1u)] = key[k];
}
__global__ void wrap_pick_kernel(u32 n_w, const u32* __restrict__ w_start, const u64* __restrict__ w_jstar, const u64* __restrict__ sorted, u64* __restrict__ ath_override) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_w) ath_override[r] = sorted[w_start[r] + w_jstar[r] - 1];
}

// number of k-min-mer occurrences of a batch: sum over its reads of (n > k ? n - k + 1 : 0)   (src/main.rs:756-759)
// the same, counting only the windows owned by `rank` (replicated-sketch mode); one thread per minimizer index
__global__ __launch_bounds__(256) void count_owned_windows_kernel(const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i0, u64 i1,
                                                                  u32 k, u32 world, const u64* thr, u32 rank, u64* __restrict__ out) {
    constexpr int WPT = 4;                   // four candidates per thread: their dependent loads overlap
    const u64 b0 = i0 + (u64)blockIdx.x * (256 * WPT);
    u32 slot[WPT]; bool ok[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) { const u64 i = b0 + u * 256 + threadIdx.x; ok[u] = i < i1; slot[u] = ok[u] ? mread[i] : 0; }
    u32 mine = 0;
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const u64 i = b0 + u * 256 + threadIdx.x;
        if (ok[u]) {
            const u64 rs = roff[slot[u]], re = roff[slot[u] + 1];
            if (re - rs > k && i + k <= re && window_owner(mh + i, k, OwnerSpec{world, thr}) == rank) ++mine;
        }
    }
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd((unsigned long long*)ctr_shard(out), (unsigned long long)mine);
}
// per-owner window counts of a batch (what a rank tells its peers, so that nobody has to re-count a foreign sketch)
__global__ __launch_bounds__(256) void owner_hist_kernel(const u64* __restrict__ mh, const u32* __restrict__ mread, const u64* __restrict__ roff, u64 i0, u64 i1,
                                                         u32 k, u32 world, const u64* thr, u64* __restrict__ counts) {
    extern __shared__ u32 hist[];
    for (u32 t = threadIdx.x; t < world; t += 256) hist[t] = 0;
    __syncthreads();
    constexpr int WPT = 4;
    const u64 b0 = i0 + (u64)blockIdx.x * (256 * WPT);
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const u64 i = b0 + u * 256 + threadIdx.x;
        if (i < i1) {
            const u32 slot = mread[i];
            const u64 rs = roff[slot], re = roff[slot + 1];
            if (re - rs > k && i + k <= re) atomicAdd(&hist[window_owner(mh + i, k, OwnerSpec{world, thr})], 1u);
        }
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < world; t += 256) if (hist[t]) atomicAdd((unsigned long long*)&counts[t], (unsigned long long)hist[t]);
}
__global__ __launch_bounds__(1024) void count_windows_kernel(const u64* __restrict__ roff, u32 slot0, u32 n_reads, u32 k, u64* __restrict__ out) {
    __shared__ u64 ws[16];
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    u64 w = 0;
    if (r < n_reads) { const u64 n = roff[slot0 + r + 1] - roff[slot0 + r]; if (n > k) w = n - k + 1; }
    for (int d = 32; d; d >>= 1) w += __shfl_down(w, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {                  // one atomic per workgroup: same-address atomics serialise (~12 ns each)
        u64 t = 0;
        for (int i = 0; i < 16; ++i) t += ws[i];
        if (t) atomicAdd((unsigned long long*)out, (unsigned long long)t);
    }
}

// One thread per solid node for the scalar fields; the k key values of a workgroup's 256 nodes are then copied by
// the whole workgroup, consecutive lanes on consecutive values, so that every row of o_keys is written as one run.
__global__ __launch_bounds__(256) void fin_emit_kernel(FinArgs F, u64 n_solid) {
    __shared__ u64 sh_src[256];              // minimizer index of the A-th sighting | reversed << 63
    __shared__ u64 sh_row[256];
    const u64 q0 = (u64)blockIdx.x * 256, q = q0 + threadIdx.x;
    const u32 k = F.k;
    if (F.n_solid_dev) { const u64 n = *F.n_solid_dev; if (n > n_solid || q0 >= n) return; n_solid = n; }      // (uniform over the workgroup)
    if (q < n_solid) {
        const u64 s = F.order ? F.order[q] : F.solid_list[q];
        const Slot e = F.tab[s];
        const SlotView v = slot_view(e, s, F.mx, F.casc, F.A, rep_ordinal(F, e.word), F.ath_override);
        u64 i1, D; decode_ordinal(F, v.first, i1, D);
        const u64 below = (1ull << (D & 63)) - 1;
        const u64 row = F.pre_solid[D >> 6] + __popcll(F.bm_solid[D >> 6] & below);          // row of the node in index order
        const u64 n = F.o_row ? q : row;
        if (F.o_row) F.o_row[q] = row;
        F.o_index[n] = F.pre_first[D >> 6] + __popcll(F.bm_first[D >> 6] & below);           // NODE_INDEX order (main.rs:661)
        F.o_abund[n] = (u16)v.count;
        // the A-th sighting (main.rs:680-684): seqlen, shift and the sequence's origin
        const u64 oa = v.ath;
        u64 i, Da; decode_ordinal(F, oa, i, Da);
        const u64* w = F.mh + i; const u32* p = F.mpos + i;
        const bool rev = window_reversed(w, k);
        sh_src[threadIdx.x] = i | ((u64)rev << 63); sh_row[threadIdx.x] = n;
        const u64 first = p[1] - p[0], last = p[k - 1] - p[k - 2];                             // main.rs:769-776
        const u64 s0 = rev ? last : first, s1 = rev ? first : last;
        F.o_seqlen[n] = (u32)((u64)p[k - 1] + 1 - p[0] + 1);                                   // main.rs:778 (read_offsets.2)
        F.o_shift[2 * n] = (u16)s0; F.o_shift[2 * n + 1] = (u16)s1;                            // main.rs:675
        F.o_shift_full[2 * n] = s0; F.o_shift_full[2 * n + 1] = s1;
        F.o_src_read[n] = oa >> WIN_BITS; F.o_src_start[n] = p[0]; F.o_src_end[n] = (u64)p[k - 1] + F.l;
        F.o_rev[n] = rev ? 1 : 0;
    }
    __syncthreads();
    // the keys: one wavefront per node, lane j copies element j (contiguous on both sides, no index arithmetic per element)
    const u32 nodes = (u32)(n_solid - q0 < 256 ? n_solid - q0 : 256);
    const u32 lane = threadIdx.x & 63;
    // (EMIT_NB nodes per wave and round, loads before stores: one node at a time was one full round trip per node, 64 in a row per wave)
    constexpr int EMIT_NB = 8;
    for (u32 g0 = (threadIdx.x >> 6) * EMIT_NB; g0 < nodes; g0 += 4 * EMIT_NB) {
        for (u32 j = lane; j < k; j += 64) {
            u64 v[EMIT_NB];
#pragma unroll
            for (int u = 0; u < EMIT_NB; ++u) {
                const u64 src = sh_src[g0 + u < nodes ? g0 + u : nodes - 1];
                v[u] = F.mh[(src & ~(1ull << 63)) + ((src >> 63) ? k - 1 - j : j)];
            }
#pragma unroll
            for (int u = 0; u < EMIT_NB; ++u) if (g0 + u < nodes) F.o_keys[sh_row[g0 + u] * k + j] = v[u];
        }
    }
}

// order-free digest of a node table (include/mdbg_hip.h, mdbg_nodes_digest): one thread per node, the workgroup's sum and XOR go to out[0], out[1] with one atomic each
__global__ __launch_bounds__(256) void nodes_digest_kernel(const u64* __restrict__ keys, const u16* __restrict__ abund, u64 n, u32 k, unsigned long long* __restrict__ out) {
    __shared__ u64 ws[2][4];
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 h = 0;
    if (q < n) {
        h = 0x243F6A8885A308D3ull ^ (u64)abund[q];
        const u64* kp = keys + q * k;
        for (u32 j = 0; j < k; ++j) h = fmix64(h ^ kp[j]);
    }
    u64 sm = h, xr = h;
    for (int d = 32; d; d >>= 1) { sm += __shfl_down(sm, d, 64); xr ^= __shfl_down(xr, d, 64); }
    if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = sm; ws[1][threadIdx.x >> 6] = xr; }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&out[0], (unsigned long long)(ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3])); atomicXor(&out[1], (unsigned long long)(ws[1][0] ^ ws[1][1] ^ ws[1][2] ^ ws[1][3])); }
}
void launch_nodes_digest(const u64* keys, const u16* abund, u64 n, u32 k, u64* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(nodes_digest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, abund, n, k, (unsigned long long*)out);
}

// ---- positions of remote sketches, fetched on demand (multi-GPU sketch exchange, include/mdbg_dist.h) ----------------------------------
// The ranks exchange HASHES only (8 of the 12 bytes per minimizer).  Raw positions are needed for one thing: seqlen / shift / origin of
// the A-th sighting of a solid node (fin_emit_kernel reads p[0], p[1], p[k-2], p[k-1] of that window), and only the rank that sketched
// the read has them.  Before fin_emit every rank lists, per sketching rank, the A-th-sighting ordinals of its solid nodes that lie in
// somebody else's reads (pos_query_kernel: count pass, then write pass), the sketching rank answers with the four positions
// (pos_answer_kernel, through its own batch tables), and the answers are written into the local position array at the window's own
// indices (pos_scatter_kernel) — fin_emit then runs unchanged.  A few MB per finalize instead of 4 bytes per minimizer per step.
struct PosQueryArgs {
    const u32* batch_src;          // [F.bt.n] sketching rank of every batch, in F.bt order (sorted by first ordinal)
    u32 me, world, pass;           // pass 0: counts[peer] += 1; pass 1: write at offs[peer] + fill[peer]++
    unsigned long long* counts; const u64* offs; unsigned long long* fill;
    u64* q_ord; u64* q_idx;        // the query (ordinal of the A-th sighting) and the local store index of that window
};
__device__ inline u32 batch_of_ordinal(const FinArgs& F, u64 ord) {
    const u64 ro = ord >> WIN_BITS;
    u32 lo = 0, hi = F.bt.n - 1;
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if (F.bt.first_ordinal[mid] <= ro) lo = mid; else hi = mid - 1; }
    return lo;
}
// (Per-peer counts and slots go through LDS: one global atomic per peer and workgroup.  One per node on `world` addresses was 6 ms per finalize at 8
// ranks — same-address atomics serialise.)
__global__ __launch_bounds__(256) void pos_query_kernel(FinArgs F, u64 n_solid, PosQueryArgs Q) {
    __shared__ u32 cnt[OWNL_MAX_WORLD];
    __shared__ u64 base[OWNL_MAX_WORLD];
    if (threadIdx.x < OWNL_MAX_WORLD) cnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 src = 0xFFFFFFFFu, local = 0; u64 ath = 0;
    if (q < n_solid) {
        const u64 s = F.solid_list[q];
        const Slot e = F.tab[s];
        const SlotView v = slot_view(e, s, F.mx, F.casc, F.A, rep_ordinal(F, e.word), F.ath_override);
        ath = v.ath;
        src = Q.batch_src[batch_of_ordinal(F, ath)];
        if (src == Q.me || src >= Q.world || src >= OWNL_MAX_WORLD) src = 0xFFFFFFFFu;
        else local = atomicAdd(&cnt[src], 1u);
    }
    __syncthreads();
    if (threadIdx.x < Q.world && threadIdx.x < OWNL_MAX_WORLD && cnt[threadIdx.x]) {
        if (Q.pass == 0) atomicAdd(&Q.counts[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
        else base[threadIdx.x] = atomicAdd(&Q.fill[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    }
    if (Q.pass == 0) return;
    __syncthreads();
    if (src == 0xFFFFFFFFu) return;
    const u64 at = Q.offs[src] + base[src] + local;
    u64 i, D; decode_ordinal(F, ath, i, D);
    Q.q_ord[at] = ath; Q.q_idx[at] = i;
}
// answers: {p[0], p[1], p[k-2], p[k-1]} of the window with the given ordinal in THIS rank's store; *bad counts ordinals that are not
// windows of a batch this rank sketched (a protocol error)
__global__ __launch_bounds__(256) void pos_answer_kernel(FinArgs F, const u64* __restrict__ ords, u64 n, const u32* __restrict__ batch_src, u32 me, const u64* __restrict__ batch_m1,
                                                         uint4* __restrict__ ans, unsigned long long* __restrict__ bad) {
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u64 ord = ords[q];
    const u32 b = batch_of_ordinal(F, ord);
    const u64 ro = ord >> WIN_BITS;
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (batch_src[b] != me || ro < F.bt.first_ordinal[b] || ro - F.bt.first_ordinal[b] >= F.bt.n_reads[b]) { atomicAdd(bad, 1ull); ans[q] = a; return; }
    u64 i, D; decode_ordinal(F, ord, i, D);
    if (i + F.k > batch_m1[b]) { atomicAdd(bad, 1ull); ans[q] = a; return; }
    const u32* p = F.mpos + i;
    ans[q] = make_uint4(p[0], p[1], p[F.k - 2], p[F.k - 1]);
}
__global__ __launch_bounds__(256) void pos_scatter_kernel(const u64* __restrict__ idx, const uint4* __restrict__ ans, u64 n, u32 k, u32* __restrict__ mpos) {
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u64 i = idx[q]; const uint4 a = ans[q];
    mpos[i] = a.x; mpos[i + 1] = a.y; mpos[i + k - 2] = a.z; mpos[i + k - 1] = a.w;      // (k = 2: the same two entries twice, same values)
}
void launch_pos_query(const FinArgs& F, u64 n_solid, const PosQueryArgs& Q, hipStream_t s) {
    if (n_solid) hipLaunchKernelGGL(pos_query_kernel, dim3((unsigned)((n_solid + 255) / 256)), dim3(256), 0, s, F, n_solid, Q);
}
void launch_pos_answer(const FinArgs& F, const u64* ords, u64 n, const u32* batch_src, u32 me, const u64* batch_m1, uint4* ans, unsigned long long* bad, hipStream_t s) {
    if (n) hipLaunchKernelGGL(pos_answer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, ords, n, batch_src, me, batch_m1, ans, bad);
}
void launch_pos_scatter(const u64* idx, const uint4* ans, u64 n, u32 k, u32* mpos, hipStream_t s) {
    if (n) hipLaunchKernelGGL(pos_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, ans, n, k, mpos);
}

// exclusive prefix of popcounts over 64-bit words, for the two finalize bitmaps at once: pre[w] = sum_{v<w} popc(bm[v]).  Block sums, scan
// of the block sums, per-word prefix — and with at most 1024 blocks (64 M bits) the last kernel adds up the sums in front of its block
// itself: two launches for both bitmaps where there were six (small kernels in a row cost ~5 us each on the device, more on the host)
__global__ __launch_bounds__(1024) void popc_block_kernel(const u64* __restrict__ bm0, const u64* __restrict__ bm1, u64 n_words, u32* __restrict__ block_sum, u32 n_blocks) {
    __shared__ u32 ws[2][16];
    const u64 w = (u64)blockIdx.x * 1024 + threadIdx.x;
    u32 v0 = w < n_words ? __popcll(bm0[w]) : 0, v1 = w < n_words ? __popcll(bm1[w]) : 0;
    for (int d = 32; d; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); }
    if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = v0; ws[1][threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x < 2) { u32 t = 0; for (int i = 0; i < 16; ++i) t += ws[threadIdx.x][i]; block_sum[threadIdx.x * n_blocks + blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void popc_scan_blocks_kernel(u32* __restrict__ block_sum_all, u32 n_blocks) {
    __shared__ u32 ws[16]; __shared__ u32 run;
    u32* const block_sum = block_sum_all + (size_t)blockIdx.x * n_blocks;        // one workgroup per bitmap
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (u32 i0 = 0; i0 < n_blocks; i0 += 1024) {
        const u32 i = i0 + tid;
        const u32 v = i < n_blocks ? block_sum[i] : 0;
        const u32 inc = wave_incl_scan(v);
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        u32 b = run, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) b += ws[q]; tot += ws[q]; }
        if (i < n_blocks) block_sum[i] = b + inc - v;
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
}
// self_base: block_sum holds the plain sums (no scan kernel ran; n_blocks <= 1024)
__global__ __launch_bounds__(1024) void popc_prefix_kernel(const u64* __restrict__ bm0, const u64* __restrict__ bm1, u64 n_words, const u32* __restrict__ block_sum, u32 n_blocks,
                                                           u32 self_base, u32* __restrict__ pre0, u32* __restrict__ pre1) {
    __shared__ u32 ws[2][16], bs[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 w = (u64)blockIdx.x * 1024 + tid;
    const u32 v0 = w < n_words ? __popcll(bm0[w]) : 0, v1 = w < n_words ? __popcll(bm1[w]) : 0;
    const u32 i0 = wave_incl_scan(v0), i1 = wave_incl_scan(v1);
    u32 s0 = 0, s1 = 0;
    if (self_base) {
        if ((u32)tid < blockIdx.x) { s0 = block_sum[tid]; s1 = block_sum[n_blocks + tid]; }
        for (int d = 32; d; d >>= 1) { s0 += __shfl_down(s0, d, 64); s1 += __shfl_down(s1, d, 64); }
    }
    if (lane == 63) { ws[0][wv] = i0; ws[1][wv] = i1; }
    if (lane == 0) { bs[0][wv] = s0; bs[1][wv] = s1; }
    __syncthreads();
    u32 b0, b1;
    if (self_base) { b0 = 0; b1 = 0; for (int q = 0; q < 16; ++q) { b0 += bs[0][q]; b1 += bs[1][q]; } }
    else { b0 = block_sum[blockIdx.x]; b1 = block_sum[n_blocks + blockIdx.x]; }
    for (int q = 0; q < wv; ++q) { b0 += ws[0][q]; b1 += ws[1][q]; }
    if (w < n_words) { pre0[w] = b0 + i0 - v0; pre1[w] = b1 + i1 - v1; }
}

// imported sketches: mread[i] = slot of the read minimizer i belongs to (one wave per read)
__global__ __launch_bounds__(256) void fill_mread_kernel(const u64* __restrict__ roff, u32 slot0, u32 n_reads, u32* __restrict__ mread) {
    const u32 r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const u64 a = roff[slot0 + r], b = roff[slot0 + r + 1];
    for (u64 i = a + (threadIdx.x & 63); i < b; i += 64) mread[i] = slot0 + r;
}
// roff[slot0 + r] = m0 + rel[r] for r in [0, n_reads]
__global__ void rebase_offsets_kernel(const u64* __restrict__ rel, u32 n_reads, u64 m0, u64* __restrict__ roff_out) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) roff_out[r] = m0 + rel[r];
}

// same for an imported region, with the caller's offsets checked on the device: [0] = 0, non-decreasing, [n_reads] = n_min
__global__ void rebase_offsets_checked_kernel(const u64* __restrict__ rel, u32 n_reads, u64 m0, u64 n_min, u64* __restrict__ roff_out, u64* __restrict__ bad) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    const u64 v = rel[r];
    if (v > n_min || (r == 0 && v != 0) || (r == n_reads && v != n_min) || (r < n_reads && rel[r + 1] < v)) *bad = 1;
    roff_out[r] = m0 + (v > n_min ? n_min : v);
}

// out[j] = sum of shard array j (CTR_SHARDS u64 each); one block per array
__global__ __launch_bounds__(256) void sum_shards_kernel(const u64* __restrict__ shards, u64* __restrict__ out) {
    __shared__ u64 ws[4];
    const u64* s = shards + (size_t)blockIdx.x * CTR_SHARDS;
    u64 v = 0;
    for (int i = threadIdx.x; i < CTR_SHARDS; i += 256) v += s[i];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// What the host reads between the stages, in one launch: scalars[idx[j]] = sum of shard array j (up to three), then all n scalars -> pinned
// host memory (was: the sum kernels, a copy kernel and the runtime's staging of a pageable destination in front of every host decision).
// host[n] = seq is written last (system-scope fence in between): the host polls that word instead of waiting for the queue's completion signal
struct PublishArgs { const u64* shards[3]; u32 idx[3]; u32 n_arrays; u64* scalars; u32 n; u32 zero_idx; u64* host; u64 seq;      // zero_idx: scalar reset once it has been published (>= n: none)
                     u64 zero_mask; u32 zero_arrays; };      // zero_mask: further scalars reset behind the copy (bit i: scalar i); zero_arrays: bit j: shard array j is zeroed once it has been summed
                                                              // (the finalize counters are left clean for the next finalize: no zeroing launch in front of it)
__global__ __launch_bounds__(1024) void publish_scalars_kernel(PublishArgs p) {
    __shared__ u64 ws[3][16];
    static_assert(CTR_SHARDS % 1024 == 0, "whole rounds");
    for (u32 j = 0; j < p.n_arrays; ++j) {
        u64 v = 0;
#pragma unroll
        for (int i = 0; i < CTR_SHARDS / 1024; ++i) v += p.shards[j][threadIdx.x + 1024 * i];
        if ((p.zero_arrays >> j) & 1u) for (int i = 0; i < CTR_SHARDS / 1024; ++i) ((u64*)p.shards[j])[threadIdx.x + 1024 * i] = 0;      // (every entry is read and zeroed by the same thread)
        for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0) ws[j][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < p.n) {
        u64 v = p.scalars[threadIdx.x];
        for (u32 j = 0; j < p.n_arrays; ++j) if (threadIdx.x == p.idx[j]) { v = 0; for (int q = 0; q < 16; ++q) v += ws[j][q]; p.scalars[threadIdx.x] = v; }
        __hip_atomic_store(p.host + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == p.zero_idx || ((p.zero_mask >> threadIdx.x) & 1ull)) p.scalars[threadIdx.x] = 0;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p.host + p.n, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish_scalars(const PublishArgs& p, hipStream_t s) {
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(1024), 0, s, p);
}

// ---- host launchers -------------------------------------------------------------------------------
void launch_sum_shards(const u64* shards, u32 n_arrays, u64* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_shards_kernel, dim3(n_arrays), dim3(256), 0, s, shards, out);
}
void launch_clear_table(Slot* tab, u64 cap, u64* mx, u64 n_mx, hipStream_t s, const ZeroList* z = nullptr) {
    ZeroList none{};
    hipLaunchKernelGGL(clear_table_kernel, dim3(2048), dim3(256), 0, s, tab, cap, mx, n_mx, z ? *z : none);
}
void launch_insert_windows(const TableArgs& T, const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 slot0,
                           u64 first_ordinal, u64* n_windows, u32* cap_err, hipStream_t s, const u64* i1_dev = nullptr, u64 n_starts = 0) {
    if (i1 <= i0) return;
    (void)n_windows;
    const u64 n = n_starts ? n_starts : i1 - i0;          // n_starts: only the window starts [i0, i0 + n_starts) (a slice; i1 stays the end of the batch)
    const size_t codes = T.own_world > 1 && T.own_thr ? 2 * ((size_t)OWN_SPAN + T.ks.k) * sizeof(u16) : 0;      // partitioned table: the owner codes of the staged hashes
    hipLaunchKernelGGL(insert_windows_kernel, dim3((unsigned)((n + OWN_SPAN - 1) / OWN_SPAN)), dim3(256),
                       (OWN_SPAN + T.ks.k) * sizeof(u64) + OWN_SPAN * sizeof(u16) + 16 + OWN_SPAN + codes, s, T, mh, mread, roff, i0, i1, slot0, first_ordinal, cap_err, i1_dev, n);
}
// Several small regions zeroed (and one scalar set) by ONE launch: the steps between the big kernels would otherwise be chains of
// 5-microsecond fill kernels (ten of them in front of the sketch, five in front of finalize).
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroList z) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride = (u64)gridDim.x * blockDim.x;
#pragma unroll
    for (int r = 0; r < 6; ++r) for (u64 i = i0; i < z.n[r]; i += stride) z.p[r][i] = 0;
    if (i0 == 0 && z.set_p) *z.set_p = z.set_v;
}
__global__ __launch_bounds__(256) void fill_u64_kernel(u64* __restrict__ p, u64 n, u64 v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_u64(u64* p, u64 n, u64 v, hipStream_t s) {
    if (n) hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)std::min<u64>(1024, (n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
void launch_zero_regions(const ZeroList& z, hipStream_t s) {
    u64 mx = 1;
    for (int r = 0; r < 6; ++r) mx = z.n[r] > mx ? z.n[r] : mx;
    const unsigned blocks = (unsigned)std::min<u64>(1024, (mx + 255) / 256);
    hipLaunchKernelGGL(zero_regions_kernel, dim3(blocks), dim3(256), 0, s, z);
}
// A batch inserted in SLICES (dense settings: hundreds of millions of windows of which few are new keys — sizing the table for all of them would make it
// tens of GB): before slice `id` the table must have room for `bound` more keys (one per window start of the slice at most).  The first slice that does
// not fit sets *too_small (which makes it and every later insert kernel of the round return at once) and leaves its id; the host grows the table and
// resumes there.
__global__ __launch_bounds__(256) void slice_check_kernel(const u64* __restrict__ distinct_shards, u64* __restrict__ n_distinct, u64 bound, u64 cap, u32* __restrict__ too_small,
                                                          u64* __restrict__ fail_at, u64 id) {
    __shared__ u64 ws[4];
    if (*too_small) return;
    u64 v = 0;
    for (int i = threadIdx.x; i < CTR_SHARDS; i += 256) v += distinct_shards[i];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 nd = ws[0] + ws[1] + ws[2] + ws[3];
        *n_distinct = nd;
        const u64 n = nd + bound;
        if (n + n / 2 + 1024 > cap) { *fail_at = id; *too_small = 1u; }
    }
}
void launch_slice_check(const u64* distinct_shards, u64* n_distinct, u64 bound, u64 cap, u32* too_small, u64* fail_at, u64 id, hipStream_t s) {
    hipLaunchKernelGGL(slice_check_kernel, dim3(1), dim3(256), 0, s, distinct_shards, n_distinct, bound, cap, too_small, fail_at, id);
}
void launch_reserve_check(const u64* distinct_shards, u64* n_distinct, const u64* batch_windows, u64 cap, u32* too_small, hipStream_t s,
                          const u64* carry = nullptr, u64 store_cap = 0, const u32* over_max = nullptr) {
    hipLaunchKernelGGL(reserve_check_kernel, dim3(1), dim3(256), 0, s, distinct_shards, n_distinct, batch_windows, cap, too_small, carry, store_cap, over_max);
}
void launch_insert_records(const TableArgs& T, u64 r0, u64 r1, u64* n_windows, hipStream_t s) {
    if (r1 <= r0) return;
    hipLaunchKernelGGL(insert_records_kernel, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, s, T, r0, r1, n_windows);
}
void launch_rehash(const Slot* old, u64 old_cap, const u64* old_mx, const TableArgs& T, hipStream_t s) {
    hipLaunchKernelGGL(rehash_kernel, dim3((unsigned)((old_cap + 255) / 256)), dim3(256), 0, s, old, old_cap, old_mx, T);
}
// out[0], out[1] = bits set in the two bitmaps (last prefix + popcount of the last word)
__global__ void bitmap_totals_kernel(const u64* __restrict__ bm0, const u32* __restrict__ pre0, const u64* __restrict__ bm1, const u32* __restrict__ pre1, u64 n_words, u64* __restrict__ out) {
    if (threadIdx.x == 0) out[0] = n_words ? (u64)pre0[n_words - 1] + (u64)__popcll(bm0[n_words - 1]) : 0;
    if (threadIdx.x == 1) out[1] = n_words ? (u64)pre1[n_words - 1] + (u64)__popcll(bm1[n_words - 1]) : 0;
}
void launch_bitmap_totals(const u64* bm0, const u32* pre0, const u64* bm1, const u32* pre1, u64 n_words, u64* out, hipStream_t s) {
    hipLaunchKernelGGL(bitmap_totals_kernel, dim3(1), dim3(64), 0, s, bm0, pre0, bm1, pre1, n_words, out);
}
// block_tmp: 2 * ceil(n_words / 1024) u32
// have_block_sums: block_tmp holds the plain per-block popcounts already (launch_bytes_to_bits wrote them with the bitmaps)
void launch_popc_prefix2(const u64* bm0, const u64* bm1, u64 n_words, u32* block_tmp, u32* pre0, u32* pre1, hipStream_t s, bool have_block_sums = false) {
    if (!n_words) return;
    const u32 nb = (u32)((n_words + 1023) / 1024);
    const u32 self_base = nb <= 1024 ? 1u : 0u;
    if (!have_block_sums) hipLaunchKernelGGL(popc_block_kernel, dim3(nb), dim3(1024), 0, s, bm0, bm1, n_words, block_tmp, nb);
    if (!self_base) hipLaunchKernelGGL(popc_scan_blocks_kernel, dim3(2), dim3(1024), 0, s, block_tmp, nb);
    hipLaunchKernelGGL(popc_prefix_kernel, dim3(nb), dim3(1024), 0, s, bm0, bm1, n_words, block_tmp, nb, self_base, pre0, pre1);
}
void launch_count_owned_windows(const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 k, u32 world, const u64* thr, u32 rank, u64* out_shards, hipStream_t s) {
    if (i1 > i0) hipLaunchKernelGGL(count_owned_windows_kernel, dim3((unsigned)((i1 - i0 + 1023) / 1024)), dim3(256), 0, s, mh, mread, roff, i0, i1, k, world, thr, rank, out_shards);
}
void launch_owner_hist(const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 k, u32 world, const u64* thr, u64* counts, hipStream_t s) {
    if (i1 > i0) hipLaunchKernelGGL(owner_hist_kernel, dim3((unsigned)((i1 - i0 + 1023) / 1024)), dim3(256), world * sizeof(u32), s, mh, mread, roff, i0, i1, k, world, thr, counts);
}
void launch_fill_mread(const u64* roff, u32 slot0, u32 n_reads, u32* mread, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(fill_mread_kernel, dim3((n_reads + 3) / 4), dim3(256), 0, s, roff, slot0, n_reads, mread);
}
void launch_rebase_offsets(const u64* rel, u32 n_reads, u64 m0, u64* roff_out, hipStream_t s) {
    hipLaunchKernelGGL(rebase_offsets_kernel, dim3((n_reads + 256) / 256), dim3(256), 0, s, rel, n_reads, m0, roff_out);
}
void launch_rebase_offsets_checked(const u64* rel, u32 n_reads, u64 m0, u64 n_min, u64* roff_out, u64* bad, hipStream_t s) {
    hipLaunchKernelGGL(rebase_offsets_checked_kernel, dim3((n_reads + 256) / 256), dim3(256), 0, s, rel, n_reads, m0, n_min, roff_out, bad);
}
void launch_count_windows(const u64* roff, u32 slot0, u32 n_reads, u32 k, u64* out, hipStream_t s) {
    if (!n_reads) return;
    hipLaunchKernelGGL(count_windows_kernel, dim3((n_reads + 1023) / 1024), dim3(1024), 0, s, roff, slot0, n_reads, k, out);
}
void launch_fin_order(const FinArgs& F, u64 n_solid, u64* order, hipStream_t s) {
    if (n_solid) hipLaunchKernelGGL(fin_order_kernel, dim3((unsigned)((n_solid + 255) / 256)), dim3(256), 0, s, F, n_solid, order);
}
void launch_fin_mark(const FinArgs& F, hipStream_t s) {
    static const bool one_pass = getenv("MDBG_FIN_ONE_PASS") != nullptr;      // (A/B switch: the round-5 kernel for the claim-map mode too)
    if (F.claims && !one_pass) { hipLaunchKernelGGL(fin_mark_claims_kernel, dim3((unsigned)((F.cap + 1024 * FIN_SPT - 1) / (1024 * FIN_SPT))), dim3(1024), 0, s, F); return; }
    hipLaunchKernelGGL(fin_mark_kernel, dim3((unsigned)((F.cap + 1024 * FIN_SPT - 1) / (1024 * FIN_SPT))), dim3(1024), 0, s, F);
}
void launch_wrap_list(Slot* tab, u64 cap, u32 A, bool all_solid, u64* w_jstar, u32* w_count, unsigned long long* counters, hipStream_t s) {
    hipLaunchKernelGGL(wrap_list_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, tab, cap, A, all_solid, w_jstar, w_count, counters);
}
void launch_wrap_scan_windows(const TableArgs& T, const u64* mh, const u32* mread, const u64* roff, u64 i0, u64 i1, u32 slot0, u64 first_ordinal,
                              const u32* w_start, u32* w_fill, u64* occ, hipStream_t s) {
    if (i1 > i0) hipLaunchKernelGGL(wrap_scan_windows_kernel, dim3((unsigned)((i1 - i0 + 255) / 256)), dim3(256), (256 + T.ks.k) * sizeof(u64), s, T, mh, mread, roff,
                                    i0, i1, slot0, first_ordinal, w_start, w_fill, occ);
}
void launch_wrap_scan_listed(const TableArgs& T, const u64* mh, const u64* roff, u64 m0, u64 m1, const u32* list, u64 n, u32 slot0, u32 n_reads, u64 first_ordinal,
                             const u32* w_start, u32* w_fill, u64* occ, hipStream_t s) {
    if (n) hipLaunchKernelGGL(wrap_scan_listed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, T, mh, roff, m0, m1, list, n, slot0, n_reads, first_ordinal, w_start, w_fill, occ);
}
// counts and prefix of the list's segments: scan_tmp[n / SEG_BLOCK + 2] <- payload index of every SEG_BLOCK-entry block's first entry, carry[0] (zero on entry) <- total,
// picks[B.n + 1] <- first payload index of every bucket and the total.  launch_seg_copy then packs / scatters with scan_tmp.
void launch_seg_prefix(const u32* list, u64 n, u32 k, const SegBuckets& B, u64* scan_tmp, u64* carry, u64* picks, hipStream_t s) {
    if (!n) return;
    const u32 nb = (u32)((n + SEG_BLOCK - 1) / SEG_BLOCK);
    hipLaunchKernelGGL(seg_sums_kernel, dim3(nb), dim3(256), 0, s, (const uint2*)list, n, k, B, scan_tmp);
    hipLaunchKernelGGL(tile_scan_top_kernel, dim3(1), dim3(256), 0, s, nb, scan_tmp, carry);
    hipLaunchKernelGGL(seg_pick_kernel, dim3(B.n + 1), dim3(256), 0, s, (const uint2*)list, n, k, B, scan_tmp, carry, picks);
}
void launch_seg_copy(const u32* list, u64 n, u32 k, const SegBuckets& B, const u64* scan_tmp, u64* store, u64* payload, u64 payload_n, bool to_store, hipStream_t s) {
    if (n) hipLaunchKernelGGL(seg_copy_kernel, dim3((unsigned)((n + SEG_BLOCK - 1) / SEG_BLOCK)), dim3(256), 0, s, (const uint2*)list, n, k, B, scan_tmp, store, payload, payload_n, to_store ? 1u : 0u);
}
void launch_wrap_scan_records(const TableArgs& T, u64 n_records, const u32* w_start, u32* w_fill, u64* occ, hipStream_t s) {
    if (n_records) hipLaunchKernelGGL(wrap_scan_records_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, T, n_records, w_start, w_fill, occ);
}
void launch_wrap_pick(u32 n_w, const u32* w_start, const u64* w_jstar, const u64* sorted, u64* ath_override, hipStream_t s) {
    if (n_w) hipLaunchKernelGGL(wrap_pick_kernel, dim3((n_w + 255) / 256), dim3(256), 0, s, n_w, w_start, w_jstar, sorted, ath_override);
}
void launch_fin_emit(const FinArgs& F, u64 n_solid, hipStream_t s) {
    if (n_solid) hipLaunchKernelGGL(fin_emit_kernel, dim3((unsigned)((n_solid + 255) / 256)), dim3(256), 0, s, F, n_solid);
}
