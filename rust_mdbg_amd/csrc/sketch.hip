// sketch.hip — HPC + ntHash + density filter, fused and bit-sliced (gfx950).
//
// Replaces Read::encode_rle + Read::extract_density (rust-mdbg src/read.rs:157-211) and the nthash crate's
// NtHashIterator for a whole batch of reads.  Design: DESIGN.md §3.1, arithmetic: bs_core.h.
//
//   sketch_bs_kernel<L>  one workgroup per tile of 32,512 raw bases (+ 256 bases of look-back), ONE pass:
//     1. load   the tile as two bit planes, 32 bases per word — either straight from the 2-bit packed input
//               (FMT_PLANES, 0.25 B/base) or converted from ASCII on the fly (FMT_ASCII, 1 B/base: v_dot4 gathers
//               the two code bits of 16 bases, v_perm validates the alphabet);
//     2. HPC    keep mask = "code differs from the previous base" (+ read starts), both planes squeezed by a
//               branch-free parallel-suffix compaction, appended to the tile's dense stream in LDS at their bit offset;
//     3. hash   bit-sliced ntHash: the top BS_B bits of the forward and reverse hashes of 32 l-mers per lane from
//               funnel-shifted boolean planes, compared with the bound as bit planes -> candidate bitmap;
//     4. exact  every candidate (a fraction ~2.5 d of the positions) is re-evaluated with the full 64-bit hashes
//               from a 4-base lookup table, mapped back to raw coordinates and to its read;
//     5. output the survivors go to the tile's slab in position order with their count; three tiny scan kernels and
//               gather_kernel then squeeze the slabs into the final arrays, which are therefore ordered by (read,
//               position) exactly like Read.transformed / minimizers_pos.  Tiles are independent of each other.
//   Tiles that hold a byte outside ACGT (N, lower case, garbage) or whose look-back window is one long
//   homopolymer take the generic exact walker inside the same kernel (slow_tile): exact, one thread per position.
#include "mdbg_dev.h"
#include "bs_core.h"

enum { FMT_ASCII = 0, FMT_PLANES = 1 };
constexpr int SCAN_GRAN_LOG = 6, SCAN_GRAN = 1 << SCAN_GRAN_LOG;      // tiles per address of SketchArgs::block_sum
struct TileRec;

struct SketchArgs {
    const u8* bases;             // FMT_ASCII: one byte per base
    const uint2* planes;         // FMT_PLANES: {plane0, plane1} per 32 bases, bit i = base i (include/mdbg_hip.h)
    u32 fmt;
    u64 n_bases;                 // positions >= n_bases do not exist
    const u64* offsets; u32 n_reads;
    const u32* bread;            // read containing the first staged base of tile t, [n_tiles + 2]
    const TileRec* recs;         // [n_tiles] (tile_rec_kernel)
    u32 n_tiles;
    u32 tile0, tile_end;         // the launch runs the tiles [tile0, tile_end): workgroup b, tile slot ts -> tile tile0 + b * TPW + ts
    Rec* slab; u32 slab_cap;     // records of the launch's i-th tile: slab[i * slab_cap ..), in position order
    u32* n_valid;                // [n_tiles] number of minimizers whose l-mer ENDS in the tile
    u32* n_scan;                 // [n_tiles] slab slots the tile used when they are NOT all valid records (dense settings: rejected candidates
                                 // stay in the slab with read = 0xFFFFFFFF and the gather squeezes them out), else 0
    u32* last_read;              // [n_tiles] read of the tile's last minimizer (LAST_NONE: the tile has none; LAST_IN_SLAB: look at the slab), so that
                                 // the gather need not chase the slab in front of it (one more dependent round trip per tile)
    u32* over_max;               // <- largest n_valid that did not fit its slab (0: none; the host then retries with larger slabs)
    unsigned long long* block_sum;   // non-null: [launch-local tile / SCAN_GRAN] += n_valid — the first level of the gather's scan, accumulated by the tiles themselves (zeroed by
                                 // tile_rec_kernel): one kernel less behind every tile launch.  SCAN_GRAN = 64 tiles per address, not the scan's 1,024: the ~1,500 tiles that
                                 // are resident together finish at 137 M tiles/s, and ONE address takes 83 M atomics/s (MI355X_MICROARCH.md, "fanin") — with 1,024 tiles per
                                 // address the tile kernel ran 10 % slower (profiles/r06_notes.md)
    const u64* t4;               // (2 << 2*BS_GS) x u64: {F, R} per 3-base group (bs_make_table)
    const u8* tile_flags;        // FMT_PLANES: nonzero = an exception falls into the tile's staged range (null: none)
    const u64* exc_pos; const u8* exc_val; u32 n_exc;
    u32* err_flag; unsigned long long* slow_total;
    u32 read_base;               // slot index of the batch's first read in the resident store
    u64 bound; u32 l; u32 hpc; u32 btop;   // btop = top BS_B bits of the bound
    u32 scheme, s;               // MDBG_SCHEME_SYNCMERS: s-mer length (0: every l-mer is a candidate); bound = floor(density * 4^l) then
    u32 force_slow;              // MDBG_FLAG_FORCE_GENERIC: every tile takes the generic exact walker
    u64* dbg;                    // diagnostic: per-tile phase timestamps [n_tiles][16] (null in production)
    u32 stop_phase;              // diagnostic (MDBG_STOP_PHASE): tiles stop after this phase and report no minimizers (0: run everything; 4 / 5: inside phase 4, after the
                                 // exact evaluation + placement / after the rank scan)
};

// largest r in [lo, hi] with off[r] <= p
__device__ inline u32 find_read(const u64* __restrict__ off, u32 lo, u32 hi, u64 p) {
    while (lo < hi) { u32 mid = lo + ((hi - lo + 1) >> 1); if (off[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}

// bread[t] = read containing the first staged base of tile t (t <= n_tiles + 1, clamped to the batch): written by tile_rec_kernel

// What a tile needs to know about the reads it touches, prepared once per batch so that the tile kernel reads ONE record instead of
// chasing bread[] -> offsets[]: the reads [rl, rh] that overlap the staged range and, when there are at most TREC_N of them, their
// starts relative to the first staged position.
constexpr int TREC_N = 7;
struct __attribute__((aligned(16))) TileRec { u32 rl, rh; int64_t start0; int32_t rel[TREC_N - 1]; int64_t first_base; };     // start0: read rl (may lie far in front); rel[i]: read rl+1+i;
                                                                                                                                 // first_base = offsets[0] (the tile kernel would otherwise chase it through a second scalar load before it can issue its loads)
static_assert(sizeof(TileRec) == 48, "three 16-byte words");
// bread[] and the records in ONE launch: thread t runs the searches for entries t and t + 2 side by side (two dependent chains of ~20
// loads each in flight together; two kernels in a row were 10 + 6 us and a launch)
// init: scalars the sketch starts from (three zeroed, one set), folded in here: one launch less in front of the tile kernel
struct SketchInit { u64* zero[4]; u64* set_p; u64 set_v; u64* zero_arr; u32 zero_arr_n; };      // zero_arr: the block sums the tiles of the first launch add their counts to (SketchArgs::block_sum)
__global__ void tile_rec_kernel(const u64* __restrict__ off, u32 n_reads, u64 n_bases, u32 n_tiles, u32* __restrict__ bread, TileRec* __restrict__ recs, SketchInit init,
                                int64_t stride, int64_t halo) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { for (int i = 0; i < 4; ++i) if (init.zero[i]) *init.zero[i] = 0; if (init.set_p) *init.set_p = init.set_v; }
    if (t < init.zero_arr_n) init.zero_arr[t] = 0;
    if (t >= n_tiles + 2) return;
    auto first_pos = [&](u32 e) -> u64 {
        int64_t p = (int64_t)e * stride - halo;
        if (p < 0) p = 0;
        if ((u64)p >= n_bases) p = n_bases ? (int64_t)n_bases - 1 : 0;
        return (u64)p;
    };
    const u64 p0 = first_pos(t), p2 = first_pos(t + 2);
    u32 lo0 = 0, hi0 = n_reads - 1, lo2 = 0, hi2 = n_reads - 1;
    while (lo0 < hi0 || lo2 < hi2) {
        const u32 m0 = lo0 + ((hi0 - lo0 + 1) >> 1), m2 = lo2 + ((hi2 - lo2 + 1) >> 1);
        const u64 v0 = off[m0], v2 = off[m2];
        if (lo0 < hi0) { if (v0 <= p0) lo0 = m0; else hi0 = m0 - 1; }
        if (lo2 < hi2) { if (v2 <= p2) lo2 = m2; else hi2 = m2 - 1; }
    }
    bread[t] = lo0;
    if (t >= n_tiles) return;
    const int64_t raw0 = (int64_t)t * stride - halo;
    TileRec r;
    r.rl = lo0; r.rh = lo2;
    r.start0 = (int64_t)off[r.rl] - raw0;
    for (int i = 0; i < TREC_N - 1; ++i) {
        int64_t v = r.rl + 1 + i <= r.rh ? (int64_t)off[r.rl + 1 + i] - raw0 : 0x7FFFFFFF;
        r.rel[i] = (int32_t)(v > 0x7FFFFFFF ? 0x7FFFFFFF : v);
    }
    r.first_base = (int64_t)off[0];
    recs[t] = r;
}

// ---- input accessors of the generic walker -------------------------------------------------------------
struct AsciiSrc {
    const u8* b;
    __device__ u8 at(u64 q) const { return b[q]; }
    // the 16 bytes at q0 (a multiple of 16) as two little-endian words; bytes past n_bases are never looked at by the caller
    __device__ void chunk16(u64 q0, u64 n_bases, u64& lo, u64& hi) const {
        if (q0 + 16 <= n_bases) { const uint4 v = *(const uint4*)(b + q0); lo = (u64)v.x | (u64)v.y << 32; hi = (u64)v.z | (u64)v.w << 32; return; }
        lo = hi = 0;
        for (int i = 0; i < 16 && q0 + i < n_bases; ++i) { const u64 c = b[q0 + i]; if (i < 8) lo |= c << (8 * i); else hi |= c << (8 * (i - 8)); }
    }
};
struct PlaneSrc {
    const uint2* w; const u64* exc_pos; const u8* exc_val; u32 n_exc;
    __device__ u8 at(u64 q) const {
        if (n_exc) {                                   // listed exception (N, lower case, ...)?
            u32 lo = 0, hi = n_exc;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (exc_pos[mid] < q) lo = mid + 1; else hi = mid; }
            if (lo < n_exc && exc_pos[lo] == q) return exc_val[lo];
        }
        const uint2 p = w[q >> 5];
        const u32 c = ((p.x >> (q & 31)) & 1u) | (((p.y >> (q & 31)) & 1u) << 1);
        return (u8)(0x47544341u >> (8 * c));           // "ACTG"[code]
    }
    __device__ void chunk16(u64 q0, u64 n_bases, u64& lo, u64& hi) const {
        lo = hi = 0;
        if (n_exc) { for (int i = 0; i < 16 && q0 + i < n_bases; ++i) { const u64 c = at(q0 + i); if (i < 8) lo |= c << (8 * i); else hi |= c << (8 * (i - 8)); } return; }
        const uint2 p = w[q0 >> 5];
        const u32 sh = (u32)(q0 & 31), p0 = p.x >> sh, p1 = p.y >> sh;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32 c = ((p0 >> i) & 1u) | (((p1 >> i) & 1u) << 1);
            const u64 ch = (0x47544341u >> (8 * c)) & 0xFFu;
            if (i < 8) lo |= ch << (8 * i); else hi |= ch << (8 * (i - 8));
        }
    }
};

// ---- exact generic walker (src/read.rs:157-174 semantics) ----------------------------------------------------
template <bool HPC, class Src>
__device__ inline bool kept_at(const Src& s, u64 rlo, u64 p) {
    if (!HPC || p == rlo) return true;
    const u8 c = s.at(p);
    return !(c == s.at(p - 1) && in_hpc_set(c));
}
// l-mer whose LAST HPC base is the run starting at p.  false: fewer than l HPC bases precede p in the read.
template <bool HPC, class Src>
__device__ inline bool walk_lmer(const Src& s, u64 rlo, u64 p, u32 l, u64& start, u64& hash) {
    u64 q = p, fh = 0, rh = 0;
    for (int j = (int)l - 1;; --j) {
        const u8 c = s.at(q);
        fh ^= rol64(nt_h_ascii(c), l - 1 - j);
        rh ^= rol64(nt_rc_ascii(c), j);
        if (j == 0) break;
        if (q == rlo) return false;
        u64 q2 = q - 1;
        if (HPC) { const u8 c2 = s.at(q2); if (in_hpc_set(c2)) while (q2 > rlo && s.at(q2 - 1) == c2) --q2; }
        q = q2;
    }
    start = q; hash = fh < rh ? fh : rh;
    return true;
}

// ---- tile state in LDS ------------------------------------------------------------------------------------
constexpr int WPT = TILE_WPT;                 // raw words per thread
constexpr int DPAD = 4;                       // zero words in front of the dense stream (look-back of the first words)
template <int NW> struct TG {                 // geometry of a tile staged by NW waves
    static constexpr int TT = TileGeo<NW>::THREADS;
    static constexpr int RW = TileGeo<NW>::RAW_WORDS;       // raw words staged per tile (halo included)
    static constexpr int HALO = TileGeo<NW>::HALO_BASES;
    static constexpr int HW = HALO / 32;                     // leading halo words
    static constexpr int STRIDE = TileGeo<NW>::STRIDE;
    static constexpr int QCAP = NW == 4 ? 704 : 176;         // candidates that fit the unordered list (evaluated in rounds of TT); more: the word-wise rounds
    static constexpr int RS_CAP = NW == 4 ? 32 : 8;          // read starts of a tile kept in LDS (more: binary search in global memory)
    static_assert(RW == WPT * TT && HW % WPT == 0 && HW % 2 == 0 && (RW - HW) % 2 == 0, "the halo is a whole number of threads; 16-byte aligned word pairs per thread");
};

template <int NW> struct __attribute__((aligned(16))) TileLds {
    typedef TG<NW> G;
    u32 dense[2 * (DPAD + G::RW + 4)];        // dense word D: planes at [2 * (DPAD + D)], [.. + 1]; during phases 1-2 the first RW
                                              // words hold the read-start bitmap (every reader zeroes what it read)
    u32 kw[G::RW];                            // keep mask of raw word w
    u16 rpre[G::RW + 8];                      // kept bases in front of raw word w; [RW] = all
    struct { u32 cand[G::RW + 8]; u16 cpre[G::RW + 8]; u16 list[G::QCAP]; u16 surv[G::TT]; } c;    // cpre doubles as the survivors' hashes (u64 x TT) before count_words
    int64_t rs0;                              // start of read rl relative to the first staged position (may lie far in front of it)
    int32_t rs_rel[G::RS_CAP];                // start of read rl + i, i >= 1, likewise (clamped to 2^31 - 1)
    u32 misc[NW == 4 ? 32 : 24];              // [0..4] scan scratch, [8] slow, [11] Hh, [16] next round, [17] list fill, [18..20] survivors per round
#ifdef MDBG_LDS_PAD
    u32 pad_experiment[MDBG_LDS_PAD / 4];     // occupancy experiments only (scratch/build_variant.sh)
#endif
};
constexpr int T3_WORDS = 2 << (2 * BS_GS);    // exact evaluation: 3-base groups {F, R}, one table per WORKGROUP (every wave writes the same values into it)
template <int SCHEME, int NW> struct TileLdsS : TileLds<NW> {};
// syncmers: + a bitmap over DENSE positions (a read starts here).  The generic machine of a flagged tile keeps its ring of s-mer hashes (32 x 128 words: it runs on
// half of the tile's threads) on top of the structure, which is dead by then — with a ring for all 256 threads the structure had to be padded to 32 KB and only four
// workgroups fitted a CU
template <int NW> struct TileLdsS<1, NW> : TileLds<NW> {
    u32 dstart[TG<NW>::RW + 8];
};
constexpr int SYNC_SLOW_THREADS = 128;       // threads of a tile that run the generic syncmer machine: their ring (32 x 128 words = 16 KB) lies on top of the tile state, dead by then
static_assert(sizeof(TileLdsS<1, 4>) >= 32 * SYNC_SLOW_THREADS * 4, "the ring of the generic syncmer machine fits the tile state");
// FMT_ASCII stages the half planes of its 16-base chunks (2 * RW words, phase 1 only) in memory that is idle then: the part of the dense
// stream behind the read-start bitmap plus the keep masks (the stream part is zeroed again before phase 2 writes it)
template <int NW> struct StageAt {
    static constexpr int RW = TG<NW>::RW;
    static constexpr int AT = 2 * (DPAD + RW + 4) + RW - 2 * RW;      // index into dense[]: the stage ends where kw[] ends
    static_assert(AT >= RW && AT % 4 == 0 && offsetof(TileLds<NW>, kw) == sizeof(u32) * 2 * (DPAD + RW + 4), "stage = dense[AT ..) + kw[]");
    static_assert((2 * (DPAD + RW + 4) - AT) == 4 * TG<NW>::TT, "one 16-byte store per thread");
};
// NW = 4: six workgroups per CU (22.8 KB each + the 1 KB table).  NW = 1: 5.9 KB per wave tile; four of them and one table per 256-lane workgroup
// (24.6 KB: six per CU = 24 waves), or one per 64-lane workgroup (6.9 KB: 22 per CU).
#ifndef MDBG_LDS_PAD
static_assert((sizeof(TileLds<4>) + T3_WORDS * 8) * 6 <= 160 * 1024, "NW = 4: 6 workgroups per CU");
static_assert((sizeof(TileLds<1>) * 4 + T3_WORDS * 8) * 6 <= 160 * 1024, "NW = 1, four tiles per workgroup: 6 workgroups per CU");
#endif
static_assert((TG<4>::RW + 8) * 2 >= TG<4>::TT * 8 && (TG<1>::RW + 8) * 2 >= TG<1>::TT * 8 && ((TG<1>::RW + 8) * 4) % 8 == 0 && offsetof(TileLds<4>, c) % 16 == 0 && offsetof(TileLds<1>, c) % 16 == 0, "cpre holds one u64 per thread");
static_assert(TG<1>::RS_CAP > TREC_N, "the read starts of a tile record fit rs_rel");

// 16 ASCII bases -> {plane0 half | plane1 half}, MSB first (base 0 in bits 31 / 15); bad != 0: a byte outside ACGT
__device__ inline u32 ascii16_to_hp(uint4 v, u32& bad) {
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 a0h = 0, a0l = 0, a1h = 0, a1l = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32 sel = w[d] & 0x06060606u;                                    // 2 * code per byte
        const u32 recon = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, sel);   // code -> 'A','C','T','G'
        bad |= recon ^ w[d];
        const u32 m0 = w[d] & 0x02020202u, m1 = w[d] & 0x04040404u;
        const u32 wt = (d & 1) ? 0x01020408u : 0x10204080u;
        if (d < 2) { a0h = __builtin_amdgcn_udot4(m0, wt, a0h, false); a1h = __builtin_amdgcn_udot4(m1, wt, a1h, false); }
        else       { a0l = __builtin_amdgcn_udot4(m0, wt, a0l, false); a1l = __builtin_amdgcn_udot4(m1, wt, a1l, false); }
    }
    const u32 t0 = (a0h << 8) + a0l;       // 2 * plane-0 half
    const u32 t1 = (a1h << 8) + a1l;       // 4 * plane-1 half
    return (t0 << 15) | (t1 >> 2);
}

// mask of the positions [a, b) of a 32-position word, MSB first (position 0 = bit 31)
__device__ inline u32 range_mask(int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > 32) b = 32;
    if (a >= b) return 0;
    const u32 from_a = 0xFFFFFFFFu >> (u32)a;
    const u32 from_b = b == 32 ? 0u : 0xFFFFFFFFu >> (u32)b;
    return from_a & ~from_b;
}

__device__ inline u32 dpp_wave_shr1(u32 x) {        // lane i <- lane i-1 (lane 0: 0)
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}

// LDS fetch-add issued by every active lane as ONE ds_add_rtn_u32.  atomicAdd() on a wave-uniform LDS address is rewritten by the
// compiler's atomic optimizer into a scalar loop over the active lanes (s_ff1 / v_readlane / v_writelane, ~9 scalar instructions per
// lane): in the filter loop that was 126 scalar instructions per step, half of the kernel's SALU count, for a same-address conflict the
// LDS resolves in a few cycles.
__device__ inline u32 lds_fetch_add(u32* p, u32 v) {
    u32 r;
    const u32 addr = (u32)(uintptr_t)(__attribute__((address_space(3))) u32*)p;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}

// what a tile hands to the gather: records in position order + their number
__device__ inline void put_rec(const SketchArgs& a, Rec* slab, u32 rank, u64 hash, u32 pos, u32 read) {
    if (rank < a.slab_cap) { Rec r; r.hash = hash; r.pos = pos; r.read = read; slab[rank] = r; }
}
constexpr u32 LAST_NONE = 0xFFFFFFFFu, LAST_IN_SLAB = 0xFFFFFFFEu;
__device__ inline void put_count(const SketchArgs& a, u32 gt, u32 n, bool last_known = false) {
    a.n_valid[gt] = n;
    if (a.block_sum && n) atomicAdd(a.block_sum + ((gt - a.tile0) >> SCAN_GRAN_LOG), (unsigned long long)n);
    if (a.last_read && !(last_known && n)) a.last_read[gt] = n ? LAST_IN_SLAB : LAST_NONE;
    if (a.n_scan) a.n_scan[gt] = 0;
    if (n > a.slab_cap) atomicMax(a.over_max, n);
}

// ---- generic exact path for one tile (inside the tile kernel) --------------------------------------------
template <bool HPC, int NW, class Src>
__device__ void slow_tile(const SketchArgs& a, const Src& src, u32 gt, Rec* slab, TileLds<NW>& S) {
    constexpr int TT = TG<NW>::TT;
    const int tid = threadIdx.x % TT;
    const u64 t_lo = (u64)gt * TG<NW>::STRIDE;
    u64 t_hi = t_lo + TG<NW>::STRIDE; if (t_hi > a.n_bases) t_hi = a.n_bases;
    const u32 rl = a.bread[gt], rh_ = a.bread[gt + 2];
    const u64 first_base = a.offsets[0];
    u32* tmp = S.misc;
    if (tid == 0) atomicAdd(a.slow_total, 1ull);
    u32 running = 0;
    for (u64 p0 = t_lo; p0 < t_hi; p0 += TT) {
        const u64 p = p0 + tid;
        u32 sel = 0; u64 hash = 0, start = 0; u32 r = 0; u64 rlo = 0;
        if (p < t_hi && p >= first_base) {
            r = find_read(a.offsets, rl, rh_, p);
            rlo = a.offsets[r];
            const u8 c = src.at(p);
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1;      // the host applies the length rule
            if (kept_at<HPC>(src, rlo, p) && walk_lmer<HPC>(src, rlo, p, a.l, start, hash) && hash <= a.bound) sel = 1;
        }
        u32 total;
        const u32 rank = tile_excl_scan<NW>(sel, tmp, total);
        if (sel) put_rec(a, slab, running + rank, hash, (u32)(start - rlo), r + a.read_base);
        running += total;
    }
    if (tid == 0) put_count(a, gt, running);
}

// ---- syncmer scheme (src/read.rs:215-352, --syncmers -s) -----------------------------------------------------------------
// An l-mer of the (homopolymer-compressed) read is kept when the smallest of its w = l-s+1 canonical s-mer hashes sits at the middle
// s-mer (index t-1, t = ceil(w/2)) and hash(canonical 2-bit l-mer) <= density * 4^l.  "Smallest" is the minimum TRACKED by the
// reference's sliding deque (update_window, read.rs:55-80): leftmost minimum of the first full window; later a new s-mer replaces it
// only when strictly smaller, and when the tracked s-mer leaves the window the window is rescanned from the back (rightmost minimum).
// With s = 4 there are only 136 canonical s-mers, so tied minima are common and the history matters.  The state machine is
// sequential, but it FORGETS: at any window whose minimum is unique the tracked position is that minimum whatever happened before.
// Two implementations inside the tile kernel (SCHEME = 1).  Fast tiles run the machine over the tile's DENSE stream in LDS (phases 1-2 are
// the density scheme's): one thread per stretch of ~94 dense positions, base codes straight from the bit planes, s-mer hashes in 32-bit
// arithmetic, the last 32 hashes in an LDS ring — it starts some distance in front of its stretch and runs until a unique-minimum
// window (or a read start) has been seen, from there on its state is the reference's; the selected window ends go to the candidate
// bitmap, and the exact phase of the density scheme (exact hash of the canonical 2-bit l-mer, raw coordinates, ranks, records) takes over.
// Tiles with bytes outside ACGT, or whose look-back window does not let every thread converge, run sync_slow_tile: the same machine
// byte by byte through the input accessor, on SYNC_SLOW_THREADS = 128 of the tile's threads, one per 254 raw positions (SEG), looking back 96, 384, ...
// positions, at most to the read start.
constexpr int SYNC_KEEP = 8;                        // records a thread of the generic machine keeps between the count and the write (more: it runs a second time).  A thread's
                                                    // segment is 254 positions since the machine moved to 128 threads: ~3 records at l=12 s=4 d=0.05, so the second run stays the exception
                                                    // for the sparse settings; dense ones (d >= 0.3) take it on most flagged tiles — flagged tiles are the rare ones (N, foreign bytes), and
                                                    // 16 records would cost 32 more VGPRs in the function that sets the kernel's register count
__device__ inline u64 sync_hash(u64 key, u64 mask) {          // src/read.rs:43-52
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}
// the same function on 32-bit values: with a mask of at most 32 bits every step only looks at the low 32 bits of its operands
__device__ inline u32 sync_hash32(u32 key, u32 mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}
__device__ inline u32 nt4_code(u8 c) {                         // src/read.rs:23-39 (SEQ_NT4_TABLE)
    switch (c) {
        case 0: case 'A': case 'a': return 0;
        case 1: case 'C': case 'c': return 1;
        case 2: case 'G': case 'g': return 2;
        case 3: case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

// dq: ring of the last <= 32 s-mer hashes (s <= 16: 32 bits) of each of the SYNC_SLOW_THREADS threads that run the machine, dq[slot][thread] (32 x 128 words = 16 KB,
// on top of the tile state, which is dead by then); sc_tmp: 8 words of scan scratch + 1 flag
template <bool HPC, class Src>
__device__ __attribute__((noinline)) void sync_slow_tile(const SketchArgs& a, const Src& src, u32 gt, Rec* slab, u32 (*dq)[SYNC_SLOW_THREADS], u32* sc_tmp) {
    constexpr int TT = SYNC_SLOW_THREADS, TILE_STRIDE = TG<4>::STRIDE;      // (the syncmer scheme runs on the 256-lane tiles; this machine on the first TT of their threads)
    constexpr u32 SEG = (TILE_STRIDE + TT - 1) / TT;           // raw positions per thread
    u32& any_over = sc_tmp[8];
    const int tid = threadIdx.x;
    if (tid == 0) atomicAdd(a.slow_total, 1ull);
    const u32 l = a.l, sm = a.s, w = l - sm + 1, t = (w + 1) / 2;
    const u64 smask = sm ? ((1ull << (2 * sm)) - 1) : 0, lmask = (1ull << (2 * l)) - 1;
    const u64 lshift = (u64)(l - 1) * 2, sshift = sm ? (u64)(sm - 1) * 2 : 0;
    const u64 first_base = a.offsets[0];
    const u64 t_lo = (u64)gt * TILE_STRIDE; u64 t_hi = t_lo + TILE_STRIDE; if (t_hi > a.n_bases) t_hi = a.n_bases;
    u64 own_lo = t_lo + (u64)tid * SEG, own_hi = own_lo + SEG;      // the tile owns the l-mers whose LAST base (its run start) lies in [t_lo, t_hi)
    if (own_lo < first_base) own_lo = first_base;
    if (own_hi > t_hi) own_hi = t_hi;
    if (tid >= TT) own_lo = own_hi = t_hi;                           // (the other threads only take part in the scan and the barriers)
    if (tid == 0) any_over = 0;
    __syncthreads();

    // one run of the state machine over my segment; WRITE: records go to the slab from rank `base`, else up to SYNC_KEEP are kept
    Rec keep[SYNC_KEEP]; u32 n_mine = 0;
    auto run = [&](bool write, u32 base) {
        n_mine = 0;
        if (own_lo >= own_hi) return;
        u32 r = find_read(a.offsets, 0, a.n_reads - 1, own_lo);
        for (u64 look = 96;; look *= 4) {
            // start: `look` positions in front of my segment, not before my first position's read (state is exact from a read start)
            u64 rlo = a.offsets[r], rhi = a.offsets[r + 1];
            u64 q = own_lo > rlo + look ? own_lo - look : rlo;
            bool conv = q == rlo;                                 // the machine's state equals the reference's
            u32 rr = r;
            u64 xl0 = 0, xl1 = 0, min_val = ~0ull; u32 xs0 = 0, xs1 = 0, lp = 0, cnt = 0, min_idx = 0, warm = 0;      // s <= 16: the s-mer fits 32 bits
            u8 prev = q > rlo ? src.at(q - 1) : 0;
            bool restart = false;
            u64 clo = 0, chi = 0, cbase = ~0ull;                  // the 16 input bytes around q (one load per 16 positions instead of one per position)
            for (; q < own_hi; ++q) {
                while (q >= rhi) { ++rr; rlo = rhi; rhi = a.offsets[rr + 1]; lp = 0; cnt = 0; xl0 = xl1 = xs0 = xs1 = 0; min_val = ~0ull; conv = true; }   // next read: reset
                if (q == own_lo && !conv) { restart = true; break; }
                if ((q & ~15ull) != cbase) { cbase = q & ~15ull; src.chunk16(cbase, a.n_bases, clo, chi); }
                const u32 bi = (u32)(q & 15);
                const u8 c = (u8)(((bi & 8) ? chi : clo) >> (8 * (bi & 7)));
                const bool kept = !HPC || q == rlo || !(c == prev && in_hpc_set(c));
                prev = c;
                if (!kept) continue;
                const u32 code = nt4_code(c);
                if (code >= 4) { lp = 0; cnt = 0; xl0 = xl1 = xs0 = xs1 = 0; min_val = ~0ull; conv = true; continue; }      // read.rs:334-341
                xl0 = (xl0 << 2 | code) & lmask; xl1 = xl1 >> 2 | (u64)(3 - code) << lshift;
                if (sm) { xs0 = (xs0 << 2 | code) & (u32)smask; xs1 = xs1 >> 2 | (3 - code) << (u32)sshift; }
                ++lp; ++warm;
                bool cand = false;
                if (sm == 0) cand = lp >= l;
                else if (lp >= sm) {
                    const u32 hs = sync_hash32(xs0 < xs1 ? xs0 : xs1, (u32)smask);
                    ++cnt;
                    dq[cnt & 31][tid] = hs;
                    if (cnt >= w) {
                        if (cnt == w) {                            // first full window: leftmost minimum (read.rs:283-289)
                            min_val = ~0ull;
                            for (u32 j = cnt - w + 1; j <= cnt; ++j) { const u32 v = dq[j & 31][tid]; if (v < min_val) { min_val = v; min_idx = j; } }
                        } else if (min_idx == cnt - w) {           // the tracked s-mer left: rescan from the back (read.rs:63-72)
                            min_val = ~0ull;
                            for (u32 j = cnt; j + w > cnt; --j) { const u32 v = dq[j & 31][tid]; if (v < min_val) { min_val = v; min_idx = j; } }
                        } else if (hs < min_val) { min_val = hs; min_idx = cnt; }
                        if (!conv && warm > l) {                   // everything in the window comes from bases behind my start: a unique minimum pins the state
                            u32 ties = 0;
                            for (u32 j = cnt - w + 1; j <= cnt; ++j) ties += dq[j & 31][tid] == (u32)min_val;
                            conv = ties == 1;
                        }
                        cand = min_idx == cnt - w + t;
                    }
                }
                if (sm == 0 && !conv && warm > l) conv = true;    // no tracked minimum in this mode: l genuine bases are all the state there is
                if (cand && q >= own_lo) {
                    const u64 hl = sync_hash(xl0 < xl1 ? xl0 : xl1, lmask);
                    if (hl <= a.bound) {
                        // raw position of the l-mer's first base: l-1 run starts back
                        u64 st = q;
                        for (u32 j = 1; j < l; ++j) {
                            u64 q2 = st - 1;
                            if (HPC) { const u8 c2 = src.at(q2); if (in_hpc_set(c2)) while (q2 > rlo && src.at(q2 - 1) == c2) --q2; }
                            st = q2;
                        }
                        Rec rec; rec.hash = hl; rec.pos = (u32)(st - rlo); rec.read = rr + a.read_base;
                        if (write) { if (base + n_mine < a.slab_cap) slab[base + n_mine] = rec; }
                        else if (n_mine < SYNC_KEEP) keep[n_mine] = rec;
                        ++n_mine;
                    }
                }
            }
            if (!restart) break;
            n_mine = 0;
        }
    };
    run(false, 0);
    if (n_mine > SYNC_KEEP) any_over = 1;
    u32 total;
    const u32 base = block_excl_scan_256(n_mine, sc_tmp, total);          // (its barriers publish any_over)
    if (!any_over) {
        for (u32 i = 0; i < n_mine; ++i) if (base + i < a.slab_cap) slab[base + i] = keep[i];
    } else run(true, base);
    if (tid == 0) put_count(a, gt, total);
}


// ---- fast tile kernel ---------------------------------------------------------------------------------------------
struct CandOut { u64 hash; u32 pos, read; };

// One tile per NW waves; TPW tiles per workgroup (TPW > 1 only with NW = 1: independent wave tiles that share the exact-hash table and
// never meet at a barrier).  (A persistent variant — workgroups looping over tiles with the next tile's words prefetched — was measured
// and dropped: the loop makes the compiler keep ~100 more values live across the phases, 3-4 instead of 6 waves per SIMD, 3.4-4.9 ms
// instead of 2.2 ms; capped to 80 registers it spills and is no better.  profiles/r02_notes.md.)
// SCHEME 0: density scheme, L = l (compile-time: every shift of the bit-sliced filter is a constant).  SCHEME 1: syncmers, L = 0 and l
// comes from the arguments (no bit-sliced filter: phase 3 is the window-minimum machine over the dense stream).
// WMAX (syncmers): the window w = l - s + 1 itself (1 .. 32): the register window of s-mer hashes and its loops are unrolled over exactly w entries
// with static register indices
// MDBG_HOT(cond, value) marks the run-time conditions that are constant on the benchmark's workload (packed input, interior tile, homopolymer
// compression, no exception, sparse density).  The product evaluates `cond`.  profiles/isa_issue.py makes a COPY of these sources in a temporary
// directory in which the macro yields `value` and the phase stamps become marker lines, and disassembles that copy: its ISA is the hot path, whose
// static instruction counts x trip counts are compared with the SQ counters.  No build flag of the product sources does that.
#define MDBG_HOT(cond, value) (cond)
template <int L, int SCHEME = 0, int WMAX = 1, int NW = 4, int TPW = 1>
__global__ __launch_bounds__(64 * NW * TPW, SCHEME ? 5 : 6) void sketch_bs_kernel(SketchArgs a) {
    typedef TG<NW> G;
    constexpr int TT = G::TT, RW = G::RW, HW = G::HW, QCAP = G::QCAP, RS_CAP = G::RS_CAP, STAGE_AT = StageAt<NW>::AT;
    static_assert(TPW == 1 || NW == 1, "several tiles per workgroup: wave tiles only (no workgroup barrier inside a tile)");
    static_assert(SCHEME == 0 || (NW == 4 && TPW == 1), "the syncmer scheme runs on the 256-lane tiles");
    __shared__ TileLdsS<SCHEME, NW> S_all[TPW];
    __shared__ __attribute__((aligned(16))) u64 S_t3[T3_WORDS];
    __shared__ u32 sync_tmp[SCHEME ? 16 : 1];       // scan scratch of the generic syncmer machine (its ring covers S)
    const u32 Lr = SCHEME ? a.l : (u32)L;           // l
    const int tslot = TPW == 1 ? 0 : (int)(threadIdx.x / TT);
    const int tid = TPW == 1 ? (int)threadIdx.x : (int)(threadIdx.x % TT), lane = tid & 63, wv = NW == 1 ? 0 : tid >> 6;
    TileLdsS<SCHEME, NW>& S = S_all[tslot];
    const int64_t nb = (int64_t)a.n_bases;
    const int64_t n_pairs = (nb + 31) >> 5;
    const bool hpc = MDBG_HOT(a.hpc != 0, true);
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32 wg = blockIdx.x * TPW + (u32)tslot, gt = a.tile0 + wg;       // the launch's wg-th tile
    if (TPW > 1 && gt >= a.tile_end) return;          // (a whole wave; nothing below waits for it)
    Rec* const slab = a.slab + (size_t)wg * a.slab_cap;
#define MDBG_STAMP(i) do { if (a.dbg && tid == 0) a.dbg[(size_t)gt * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
    MDBG_STAMP(0);
    const int64_t raw0 = (int64_t)gt * G::STRIDE - G::HALO;      // first staged raw position (negative for tile 0)
    const bool interior = MDBG_HOT(raw0 >= 0 && raw0 + RW * 32 <= nb, true);
    const TileRec* const rec = a.recs + gt;
    // (the record is read with vector loads — the compiler cannot prove it read-only —, so what is the same in every lane is moved to scalar
    // registers by hand: the conditions below then branch on SCC instead of travelling through the phases as lane masks)
    auto uniform64 = [](int64_t v) -> int64_t {
        return (int64_t)((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)v >> 32)) << 32 | (u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)v));
    };
    const int64_t first_base = uniform64(rec->first_base);       // positions in front of it belong to no read

    // ---- phase 1: load (issued first: everything below hides under its latency), read starts, planes -------------------
    u32 x0[WPT], x1[WPT], pv0 = 0, pv1 = 0;          // my raw words (MSB first); pv*: bit 0 = the base in front of them
    constexpr int CPT = RW * 2 / TT;                  // FMT_ASCII: 16-base chunks per thread
    uint4 av[CPT]; uint2 pr[WPT];
    const int64_t pi0 = raw0 / 32 + (int64_t)WPT * tid;       // FMT_PLANES: my first word pair (raw0 is a multiple of 32, also when negative)
    if (MDBG_HOT(a.fmt == FMT_ASCII, false)) {
        if (interior) {
            const u32x4* src = (const u32x4*)(a.bases + raw0);
#pragma unroll
            for (int g = 0; g < CPT; ++g) { const u32x4 q = __builtin_nontemporal_load(src + tid + TT * g); av[g] = make_uint4(q.x, q.y, q.z, q.w); }
        }
    } else {
        if (interior) {
            const u32x4* src = (const u32x4*)(a.planes + pi0);       // two 16-byte streaming loads
#pragma unroll
            for (int i = 0; i < WPT / 2; ++i) { const u32x4 q = __builtin_nontemporal_load(src + i); pr[2 * i] = make_uint2(q.x, q.y); pr[2 * i + 1] = make_uint2(q.z, q.w); }
        } else {
#pragma unroll
            for (int i = 0; i < WPT; ++i) { const int64_t pi = pi0 + i; pr[i] = (pi >= 0 && pi < n_pairs) ? a.planes[pi] : make_uint2(0u, 0u); }
        }
        if (tid && pi0 - 1 >= 0 && pi0 - 1 < n_pairs) { const uint2 q = a.planes[pi0 - 1]; pv0 = q.x >> 31; pv1 = q.y >> 31; }
    }
    const u32 rl = (u32)__builtin_amdgcn_readfirstlane((int)rec->rl), rh_ = (u32)__builtin_amdgcn_readfirstlane((int)rec->rh);
    static_assert((2 * (DPAD + RW + 4)) % 4 == 0, "the dense stream is a whole number of 16-byte words");
    for (int i = tid; i < 2 * (DPAD + RW + 4) / 4; i += TT) ((uint4*)S.dense)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (SCHEME == 0) {                                // (TPW > 1: every wave writes the whole table — the same values — so none waits for another)
        static_assert(T3_WORDS % 128 == 0, "the table is copied 16 bytes per lane");
        for (int i = tid; i < T3_WORDS / 2; i += TT) ((uint4*)S_t3)[i] = ((const uint4*)a.t4)[i];
    } else {                                          // 8 bits -> 16 bits, bit i to bit 2 i: the exact phase interleaves the code planes with it
        u32 v = 0;
        for (int i = 0; i < 8; ++i) v |= ((u32)tid >> i & 1u) << (2 * i);
        ((u16*)S_t3)[tid] = (u16)v;
    }
    if (tid == 0) { S.misc[8] = a.force_slow | ((a.tile_flags && a.tile_flags[gt]) ? 1u : 0u); S.misc[9] = 0; S.misc[11] = 0; S.misc[17] = 0; S.misc[18] = 0; S.misc[19] = 0; S.misc[20] = 0; }
    tile_sync<NW>();
    if (MDBG_HOT(rh_ - rl < (u32)TREC_N, true)) {                       // the usual case: the read starts come with the tile's record
        if ((u32)tid <= rh_ - rl) {
            const int64_t rel = tid == 0 ? rec->start0 : (int64_t)rec->rel[tid - 1];
            if (tid == 0) S.rs0 = rel; else S.rs_rel[tid] = (int32_t)rel;
            if (rel >= 0 && rel < RW * 32) atomicOr(&S.dense[rel >> 5], 0x80000000u >> (rel & 31));
        }
    } else for (u32 r = rl + tid; r <= rh_ && r < a.n_reads; r += TT) {
        const int64_t rel = (int64_t)a.offsets[r] - raw0;
        if (r == rl) S.rs0 = rel; else if (r - rl < RS_CAP) S.rs_rel[r - rl] = (int32_t)(rel > 0x7FFFFFFF ? 0x7FFFFFFF : rel);
        if (rel >= 0 && rel < RW * 32) atomicOr(&S.dense[rel >> 5], 0x80000000u >> (rel & 31));
    }
    u32* const stage = S.dense + STAGE_AT;
    if (MDBG_HOT(a.fmt == FMT_ASCII, false)) {
        u32 bad_any = 0;
        if (interior) {
#pragma unroll
            for (int g = 0; g < CPT; ++g) stage[tid + TT * g] = ascii16_to_hp(av[g], bad_any);
        } else {
#pragma unroll 1
            for (int g = 0; g < CPT; ++g) {
                const int ci = tid + TT * g;
                const int64_t pos = raw0 + 16 * (int64_t)ci;
                u32 w4[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};     // positions outside the batch read as 'A' (masked below)
                if (pos >= 0 && pos + 16 <= nb) { const uint4 q = *(const uint4*)(a.bases + pos); w4[0] = q.x; w4[1] = q.y; w4[2] = q.z; w4[3] = q.w; }
                else for (int i = 0; i < 16; ++i) if (pos + i >= 0 && pos + i < nb) w4[i >> 2] = (w4[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((u32)a.bases[pos + i] << (8 * (i & 3)));
                stage[ci] = ascii16_to_hp(make_uint4(w4[0], w4[1], w4[2], w4[3]), bad_any);
            }
        }
        if (bad_any) {                               // some byte of my chunks is not one of ACGT: the tile takes the generic path
            S.misc[8] = 1;
            for (int g = 0; g < CPT; ++g) {
                const int64_t pos = raw0 + 16 * (int64_t)(tid + TT * g);
                for (int i = 0; i < 16; ++i) {
                    const int64_t q = pos + i;
                    if (SCHEME == 0 && q >= 0 && q < nb) { const u8 c = a.bases[q]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1; }      // (syncmers: any other byte just resets the machine, src/read.rs:334-341)
                }
            }
        }
        tile_sync<NW>();
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const uint2 h = *(const uint2*)(stage + 2 * (WPT * tid + i));
            x0[i] = (h.x & 0xFFFF0000u) | (h.y >> 16); x1[i] = (h.x << 16) | (h.y & 0xFFFFu);
        }
        if (tid) { const u32 hp = stage[2 * WPT * tid - 1]; pv0 = hp >> 16; pv1 = hp; }
        tile_sync<NW>();                                     // the stage has been read: its stream part goes back to zero (the barriers of the scan below order this before the stream writes)
        static_assert((2 * (DPAD + RW + 4) - STAGE_AT) == 4 * TT && STAGE_AT % 4 == 0, "one 16-byte store per thread");
        ((uint4*)(S.dense + STAGE_AT))[tid] = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
        for (int i = 0; i < WPT; ++i) { x0[i] = __brev(pr[i].x); x1[i] = __brev(pr[i].y); }
        tile_sync<NW>();
    }
    MDBG_STAMP(1);
    if (MDBG_HOT(a.stop_phase == 1, false)) { if (tid == 0) a.n_valid[gt] = x0[0] == 0x12345u; return; }

    // ---- phase 2: keep masks, compaction, dense stream ----------------------------------------------------------------
    u32 kw[WPT], n_kept[WPT], mine = 0;
    {
        const int64_t lo = first_base - raw0, hi = nb - raw0;        // existing positions, tile-relative
        if (MDBG_HOT(interior && lo <= 0 && hpc, true)) {
            // the usual tile (every staged position exists, homopolymer compression on) on a path of its own: decided once, in scalar registers —
            // folded into the general loop below the compiler carried both conditions through every word as lane masks
            const uint4 st = *(const uint4*)(S.dense + WPT * tid);          // read-start bits of my four words
            *(uint4*)(S.dense + WPT * tid) = make_uint4(0u, 0u, 0u, 0u);    // consumed: back to an empty dense stream
            const u32 sb[WPT] = {st.x, st.y, st.z, st.w};
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const u32 d0 = bs_alignbit(i ? x0[i - 1] : pv0, x0[i], 1), d1 = bs_alignbit(i ? x1[i - 1] : pv1, x1[i], 1);
                u32 k = (x0[i] ^ d0) | (x1[i] ^ d1) | sb[i];
                if (i == 0) k |= tid == 0 ? 0x80000000u : 0u;     // nothing staged in front of the tile's first position
                kw[i] = k; n_kept[i] = bs_popc(k); mine += n_kept[i];
            }
        } else {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {                      // keep masks first: they look at the neighbouring raw word
            const int w = WPT * tid + i;
            const u32 vm = interior && lo <= 0 ? 0xFFFFFFFFu : range_mask(lo - 32 * (int64_t)w, hi - 32 * (int64_t)w);
            u32 k = vm;
            if (hpc) {
                const u32 d0 = bs_alignbit(i ? x0[i - 1] : pv0, x0[i], 1), d1 = bs_alignbit(i ? x1[i - 1] : pv1, x1[i], 1);
                k = ((x0[i] ^ d0) | (x1[i] ^ d1) | S.dense[w]) & vm;
                if (w == 0) k |= 0x80000000u & vm;           // nothing staged in front of the first position
            }
            S.dense[w] = 0;                                  // the read-start bitmap has been consumed: back to an empty dense stream
            kw[i] = k; n_kept[i] = bs_popc(k); mine += n_kept[i];
        }
        }
        if (hpc) {                                           // (wave-uniform: no exec juggling per word)
#pragma unroll
            for (int i = 0; i < WPT; ++i) bs_compress2(kw[i], x0[i], x1[i]);
        } else {
#pragma unroll
            for (int i = 0; i < WPT; ++i) if (kw[i] != 0xFFFFFFFFu) bs_compress2(kw[i], x0[i], x1[i]);
        }
    }
    u32 H;
    u32 off = tile_excl_scan<NW>(mine, S.misc, H);          // (its barriers also order the bitmap reset before the stream writes)
    static_assert(HW % WPT == 0, "the halo is a whole number of threads");
    if (tid == HW / WPT) S.misc[11] = off;                   // kept bases of the halo words
    if (tid == TT - 1) S.rpre[RW] = (u16)H;
    static_assert(WPT == 4, "one 16-byte store of keep masks and one 8-byte store of prefixes per thread");
    *(uint4*)(S.kw + WPT * tid) = make_uint4(kw[0], kw[1], kw[2], kw[3]);
    {
        const u32 o1 = off + n_kept[0], o2 = o1 + n_kept[1], o3 = o2 + n_kept[2];
        *(uint2*)(S.rpre + WPT * tid) = make_uint2(off | o1 << 16, o2 | o3 << 16);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const u32 n = n_kept[i];
        {   // no branches: an empty word ORs zeros, a word that does not straddle ORs zeros into the next one (the stream has spare words behind its end).
            // With `if (n)` / `if (s + n > 32)` around the two atomics the tile kernel was 2 % slower (profiles/r04_l_micro_ab.txt): the second is taken by
            // ~70 % of the lanes, so every wave ran both sides anyway and paid the exec-mask bookkeeping four times per lane.
            const u32 wi = off >> 5, s = off & 31;
            unsigned long long* dst = (unsigned long long*)(S.dense + 2 * (DPAD + wi));
            atomicOr(dst, (unsigned long long)(x0[i] >> s) | ((unsigned long long)(x1[i] >> s) << 32));
            atomicOr(dst + 1, (unsigned long long)bs_alignbit(x0[i], 0u, s) | ((unsigned long long)bs_alignbit(x1[i], 0u, s) << 32));
        }
        off += n;
    }
    tile_sync<NW>();
    MDBG_STAMP(2);
    if (MDBG_HOT(a.stop_phase == 2, false)) { if (tid == 0) a.n_valid[gt] = 0; return; }
    const u32 Hh = S.misc[11];
    const bool true_start = raw0 <= first_base;       // the stream begins inside this tile: nothing to look back at
    auto run_slow_tile = [&]() {
        // N / foreign bytes, or the look-back window is too short (one long homopolymer): exact generic path for the whole tile
        if constexpr (SCHEME == 0) {
            if (a.fmt == FMT_ASCII) { AsciiSrc src{a.bases}; if (hpc) slow_tile<true>(a, src, gt, slab, S); else slow_tile<false>(a, src, gt, slab, S); }
            else { PlaneSrc src{a.planes, a.exc_pos, a.exc_val, a.tile_flags && a.tile_flags[gt] ? a.n_exc : 0u}; if (hpc) slow_tile<true>(a, src, gt, slab, S); else slow_tile<false>(a, src, gt, slab, S); }
        } else {
            if (a.fmt == FMT_ASCII) { AsciiSrc src{a.bases}; if (hpc) sync_slow_tile<true>(a, src, gt, slab, (u32(*)[SYNC_SLOW_THREADS])&S, sync_tmp); else sync_slow_tile<false>(a, src, gt, slab, (u32(*)[SYNC_SLOW_THREADS])&S, sync_tmp); }
            else { PlaneSrc src{a.planes, a.exc_pos, a.exc_val, a.tile_flags && a.tile_flags[gt] ? a.n_exc : 0u}; if (hpc) sync_slow_tile<true>(a, src, gt, slab, (u32(*)[SYNC_SLOW_THREADS])&S, sync_tmp); else sync_slow_tile<false>(a, src, gt, slab, (u32(*)[SYNC_SLOW_THREADS])&S, sync_tmp); }
        }
    };
    if (MDBG_HOT(S.misc[8] || (!true_start && Hh < Lr), false)) { tile_sync<NW>(); run_slow_tile(); return; }

    // ---- phase 3: bit-sliced filter over the dense stream -> candidate bitmap + (unordered) candidate list ---------------
    // candidate plane coordinate x = e + BS_B - 1; owned END positions e in [max(Hh, L-1), H)
    const u32 e_lo = Hh > Lr - 1 ? Hh : Lr - 1;
    const u32 n_out = H ? ((H + BS_B - 2) >> 5) + 1 : 0;              // words of the candidate plane
    const u32 n_rs = rh_ - rl + 1;                                     // reads that touch the staged range
    if constexpr (SCHEME == 0) {
        u32 bmask[BS_B];
#pragma unroll
        for (int i = 0; i < BS_B; ++i) bmask[i] = ((a.btop >> (BS_B - 1 - i)) & 1u) ? 0xFFFFFFFFu : 0u;
        const u32 n_steps = (n_out + 62) / 63;
        const bool zero_test = MDBG_HOT(a.btop == 0, true);                          // density < 2^-BS_B: "all evaluated hash bits are zero"
        const u32 wv_s = (u32)__builtin_amdgcn_readfirstlane(wv);    // the step index lives in scalar registers
        const int nb_addr = 4 * ((lane + 63) & 63);                  // ds_bpermute address of the lane in front
        // first / last candidate-plane position that is owned: words wholly inside [x_lo, x_hi) need no mask
        const int x_lo = __builtin_amdgcn_readfirstlane((int)e_lo + BS_B - 1), x_hi = __builtin_amdgcn_readfirstlane((int)H + BS_B - 1);
        for (u32 st = wv_s; st < n_steps; st += TT / 64) {
            const int D = (int)(63 * st) + lane - 1;                  // lane 0 recomputes the word before the step's first
            const u32* dw = S.dense + 2 * (DPAD + (D < RW + 3 ? D : RW + 3));     // words past the stream are zero; their results are dropped
            // the planes of the word in front live in the lane in front.  They come through the LDS crossbar (ds_bpermute: no memory
            // access, and not a VALU slot — the kernel is bound by VALU issue; round 2 used 14 v_mov_dpp wave_shr:1 per word here)
            const uint2 c = *(const uint2*)dw, p = *(const uint2*)(dw - 2);
            uint2 q = make_uint2(0u, 0u);
            if (L + BS_B - 2 >= 32) q = *(const uint2*)(dw - 4);
            u32 W[BS_B], Wp[BS_B], inv;
            bs_strand_planes<L, true>(c.x, c.y, p.x, p.y, q.x, q.y, W, inv);
#pragma unroll
            for (int i = 0; i < BS_B - 1; ++i) Wp[i] = (u32)__builtin_amdgcn_ds_bpermute(nb_addr, (int)W[i]);
            Wp[BS_B - 1] = 0;
            u32 cand = zero_test ? bs_strand_compare<true, true>(W, Wp, inv, bmask) : bs_strand_compare<true, false>(W, Wp, inv, bmask);
            bs_strand_planes<L, false>(c.x, c.y, p.x, p.y, q.x, q.y, W, inv);
            Wp[0] = 0;
#pragma unroll
            for (int i = 1; i < BS_B; ++i) Wp[i] = (u32)__builtin_amdgcn_ds_bpermute(nb_addr, (int)W[i]);
            cand |= zero_test ? bs_strand_compare<false, true>(W, Wp, inv, bmask) : bs_strand_compare<false, false>(W, Wp, inv, bmask);
            // only the step that holds the first owned position and the one that holds the last need the range mask (wave-uniform test).
            // (Two copies of the loop, one per kind of comparison, so that no step branches on zero_test: no difference, profiles/r04_l_micro_ab.txt.)
            const int Dw0 = (int)(63 * st) - 1;
            if (32 * Dw0 < x_lo || 32 * (Dw0 + 64) > x_hi) cand &= range_mask((int64_t)x_lo - 32 * (int64_t)D, (int64_t)x_hi - 32 * (int64_t)D);
            if (lane && (u32)D < n_out) S.c.cand[D] = cand;
        }
        for (u32 D = n_out + tid; D < RW + 8; D += TT) S.c.cand[D] = 0;
    } else {
        // ---- phase 3, syncmer scheme: the window-minimum machine (src/read.rs:215-352) over the dense stream -----------------------
        const u32 l = a.l, sm = a.s, w = l - sm + 1, t = (w + 1) / 2;
        const u32 smask = sm ? (sm >= 16 ? 0xFFFFFFFFu : (1u << (2 * sm)) - 1u) : 0u, sshift = sm ? 2 * (sm - 1) : 0;
        for (int i = tid; i < RW + 8; i += TT) { S.dstart[i] = 0; S.c.cand[i] = 0; }
        tile_sync<NW>();
        // read starts in DENSE coordinates (a start is a forced run start: its dense index is the number of kept bases in front of it)
        auto mark_start = [&](int64_t rel) {
            if (rel < 0 || rel >= (int64_t)RW * 32) return;
            const u32 rw = (u32)rel >> 5, rb = (u32)rel & 31;
            const u32 dp = S.rpre[rw] + (rb ? bs_popc(S.kw[rw] >> (32 - rb)) : 0u);
            if (dp < H && ((S.kw[rw] << rb) & 0x80000000u)) atomicOr(&S.dstart[dp >> 5], 0x80000000u >> (dp & 31));
        };
        if (n_rs <= RS_CAP) { for (u32 i = tid; i < n_rs; i += TT) mark_start(i ? (int64_t)S.rs_rel[i] : S.rs0); }
        else for (u32 r = rl + tid; r <= rh_ && r < a.n_reads; r += TT) mark_start((int64_t)a.offsets[r] - raw0);
        tile_sync<NW>();
        // The machine as a SCAN (round 6; until round 5 one thread ran the reference's machine over a stretch of ~94 positions after 32 positions of
        // look-back: ~280 VALU per window).  The tracked s-mer of a window is always an occurrence of the window's smallest hash; which one, among
        // equal ones, is history: T(p) = T(p - 1) while that s-mer is still inside the window, else the RIGHTMOST occurrence (the rescan from the back);
        // a window whose minimum is UNIQUE pins T whatever happened before (a strictly smaller arrival is such a window), and so does a read's first
        // full window (the LEFTMOST occurrence).  So T(p) is a function of T(p - 1) that is CONSTANT at every anchor (unique minimum, first window) and
        // at every window without a full l-mer behind it — with s = 4 nine windows in ten are anchors.  A lane takes G consecutive windows: it builds
        // the w + G - 1 s-mer hashes it needs from the bit planes itself (registers, nothing staged), finds every window's rightmost and leftmost minimum
        // (one unsigned min each over hash << 6 | index), runs the G steps once without a predecessor, and takes its predecessor's result from the lane
        // below (a lane without an anchor — rare — waits for it: a fixed-point loop over the wave, usually zero rounds).  A wave owns a contiguous
        // quarter of the tile's positions and walks it in rounds of 64 G positions, the carry from round to round in a scalar; one look-back round in
        // front of the quarter warms the carry up.  A window of an owned position whose state is still unknown then (no anchor in 512 positions: a
        // tandem repeat) sends the tile to the generic machine.
#ifndef MDBG_SYNC_SG
#define MDBG_SYNC_SG 16
#endif
        constexpr int SG = MDBG_SYNC_SG;                   // windows per lane and round (8 or 16: a lane's view is 64 positions: l - 1 + SG of them are used)
        constexpr int NH = WMAX + SG - 1;                  // s-mer hashes of a lane: END positions p0 - (w - 1) .. p0 + SG - 1
        // s <= 4: the s-mer hash comes from a table (second half of t3: 256 x u16, filled here) indexed by the s bits of plane 1 and the s bits of plane 0 ^ plane 1
        // as they lie in the stream — the table's builder de-interleaves the index, forms the s-mer and its reverse complement and hashes the smaller: a hash costs two
        // bit-field extracts, a shift-or and a 2-byte LDS read, no rolling registers
        const bool use_lut = sm != 0 && sm <= 4;
        u16* const lut = (u16*)S_t3 + 256;
        u32* const lut32 = (u32*)S.c.list;                // 256 words on top of the candidate list, which is written behind this phase
        static_assert(sizeof(S.c.list) >= 256 * 4 && offsetof(TileLds<NW>, c) % 4 == 0 && (sizeof(S.c.cand) + sizeof(S.c.cpre)) % 4 == 0, "the packed s-mer table fits the candidate list");
        if (use_lut) {
            const u32 ia = (u32)tid >> sm, ib = (u32)tid & ((1u << sm) - 1u);
            u32 fw = 0, rc = 0;
            for (u32 j = 0; j < sm; ++j) {                 // base j of the s-mer (0 = first): its code's two bits are bit (s - 1 - j) of either half of the index
                const u32 c = ((ia >> (sm - 1 - j)) & 1u) << 1 | ((ib >> (sm - 1 - j)) & 1u);
                fw |= c << (2 * (sm - 1 - j)); rc |= (3u - c) << (2 * j);
            }
            const u32 hv = sync_hash32(fw < rc ? fw : rc, smask);
            lut[tid] = (u16)hv;
            lut32[tid] = (hv << 6) * 0x10001u;             // both halves hash << 6 (at most 14 bits: s <= 4): the packed form of the common rounds below
        }
        tile_sync<NW>();
        {
            const u32 n_own = H > e_lo ? H - e_lo : 0u;
            const u32 Qw = (((n_own + NW - 1) / NW) + (64u * SG - 1u)) & ~(64u * SG - 1u);      // owned positions per wave: whole rounds
            const u32 wv_s = (u32)__builtin_amdgcn_readfirstlane(wv);
            const u32 Awv = e_lo + wv_s * Qw, Bwv = Awv + Qw < H ? Awv + Qw : H;                   // the wave's owned positions [Awv, Bwv)
            int32_t carry = 4096;                           // (UNKI below: nothing known in front of the look-back round)
            bool need_slow = false;
            auto word64 = [](u32 x, u32 y, u32 z, u32 r) -> u64 {      // 64 positions from bit r of word x on (MSB first)
                const u32 hi = r ? (x << r) | (y >> (32u - r)) : x, lo = r ? (y << r) | (z >> (32u - r)) : y;
                return (u64)hi << 32 | lo;
            };
            const u32 fm = sm ? (1u << sm) - 1u : 0u;
            // bit jj of bits: window p0 + jj is a candidate -> candidate plane (coordinate x = p + BS_B - 1, MSB first)
            auto put_cands = [&](u32 bits, int32_t p0_) {
                if (!bits) return;
                const u32 x0c = (u32)p0_ + BS_B - 1;
                const u64 cb = ((u64)__brev(bits) << 32) >> (x0c & 31u);
                if ((u32)(cb >> 32)) atomicOr(&S.c.cand[x0c >> 5], (u32)(cb >> 32));
                if ((u32)cb) atomicOr(&S.c.cand[(x0c >> 5) + 1], (u32)cb);
            };
            if (Awv < H) for (int32_t rd = -1; (int64_t)Awv + (int64_t)rd * (64 * SG) < (int64_t)Bwv; ++rd) {
                const int32_t p0 = (int32_t)Awv + rd * (64 * SG) + lane * SG;       // my windows end at p0 .. p0 + SG - 1
                const int32_t qb = p0 - (int32_t)(l - 1);                            // the first base I look at = view index 0 (bit 63)
                const int32_t wi = qb >> 5; const u32 rr = (u32)qb & 31u;           // (arithmetic shift: floor, also in front of the stream)
                u32 pl1[3], pl0[3], stw[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {              // (clamped indices, results masked: six loads in flight, no branches)
                    const int32_t D = wi + j;
                    const int32_t Dc = D < -DPAD ? -DPAD : D > RW + 3 ? RW + 3 : D, Ds = D < 0 ? 0 : D > RW + 3 ? RW + 3 : D;
                    const uint2 dw = *(const uint2*)(S.dense + 2 * (DPAD + Dc));
                    const u32 st = S.dstart[Ds];
                    pl1[j] = D == Dc ? dw.y : 0u; pl0[j] = D == Dc ? dw.x ^ dw.y : 0u;      // the reference's codes (A 0, C 1, G 2, T 3) = (plane 1, plane 0 ^ plane 1)
                    stw[j] = D == Ds ? st : 0u;
                }
                const u64 W1 = word64(pl1[0], pl1[1], pl1[2], rr), W0 = word64(pl0[0], pl0[1], pl0[2], rr), WS = word64(stw[0], stw[1], stw[2], rr);
                // which of my windows hold a full l-mer of one read (every base of it in the stream, no read start behind its first base), and which are a read's first:
                // mask arithmetic on the view — the masks themselves are the same in every lane
                const int32_t n_mine = (int32_t)Bwv - p0;   // my windows that are owned positions of this wave: the first n_mine
                const u32 own = n_mine >= SG ? (1u << SG) - 1u : n_mine <= 0 ? 0u : (1u << n_mine) - 1u;
                const int32_t i_hi = (int32_t)H - qb;      // view indices [max(0, -qb), i_hi) are positions of the stream
                const u64 LIVE = (qb <= -64 ? 0ull : qb < 0 ? ~0ull >> (u32)(-qb) : ~0ull) & (i_hi >= 64 ? ~0ull : i_hi <= 0 ? 0ull : ~(~0ull >> (u32)i_hi));
                // (AND over runs of l positions by doubling: view index j of andrun(V, n) = V[j] & ... & V[j + n - 1])
                auto andrun = [](u64 V, u32 n) -> u64 {
                    if (n == 0) return ~0ull;
                    u32 pw = 1;
                    for (; 2 * pw <= n; pw *= 2) V &= V << pw;
                    return V & (V << (n - pw));
                };
                u32 vmask = (1u << SG) - 1u, fmask = 0;   // (nearly every round: no edge of the stream and no read start in any lane's view)
                const bool rare = __any((int)(LIVE != ~0ull || WS != 0ull)) != 0;      // an edge of the stream or a read start in some lane's view
                if (rare) {
                    const u64 OKW = andrun(LIVE, l) & andrun(~(WS << 1), l - 1);       // every base in the stream, no read start behind the first one
                    vmask = __brev((u32)(OKW >> 32)) & ((1u << SG) - 1u);              // bit jj: window jj (view index jj = bit 63 - jj)
                    fmask = vmask & __brev((u32)(WS >> 32));
                }
                if (sm == 0) {
                    // no tracked minimum: l genuine bases are all the state there is (src/read.rs:319-333)
                    if (rd >= 0) put_cands(vmask & own, p0);
                    continue;
                }
                if (use_lut && !rare) {
                    // The common round (s <= 4; every window of every lane holds a full l-mer, no read starts): both minima at once.  hash << 6 | index fits 16 bits, so a
                    // register holds the rightmost encoding (63 - index) in its high half and the leftmost one in its low half and v_pk_min_u16 serves both; the table
                    // delivers the hash in both halves.  thr[jj]: the tracked s-mer is replaced at window jj when its index is smaller — jj itself (it has left), or
                    // "any" at a window with a unique minimum.
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    auto pkmin = [](u32 x, u32 y) -> u32 { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y))); };
                    constexpr int32_t ANY = 1 << 20;
                    const u64 Z1 = W1 >> (14u - sm), Z0 = W0 >> (14u - sm);      // the s-mer ending at index ii: its last bit at bit 50 - ii (a word index: 4 x the table index)
                    u32 e[NH];
#pragma unroll
                    for (int ii = 0; ii < NH; ++ii) {
                        const u32 ka = (u32)(Z1 >> (48 - ii)) & (fm << 2), kb = (u32)(Z0 >> (48 - ii)) & (fm << 2);
                        e[ii] = *(const u32*)((const char*)lut32 + (ka << sm | kb)) | ((u32)(63 - ii) << 16 | (u32)ii);
                    }
                    u32 m[SG];
                    if constexpr (WMAX > 2) {
                        constexpr int B = WMAX - 1;       // blocks of w - 1 indices: a window is the tail of one block and the head of the next, down to its last index
                        u32 sx[NH], px[NH];
#pragma unroll
                        for (int i = NH - 1; i >= 0; --i) sx[i] = (i % B == B - 1 || i == NH - 1) ? e[i] : pkmin(e[i], sx[(i + 1) % NH]);
#pragma unroll
                        for (int i = 0; i < NH; ++i) px[i] = i % B == 0 ? e[i] : pkmin(e[i], px[(i + NH - 1) % NH]);
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) m[jj] = pkmin(sx[jj], px[jj + B]);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) { u32 x = e[jj]; for (int d = 1; d < WMAX; ++d) x = pkmin(x, e[jj + d]); m[jj] = x; }
                    }
                    int32_t val[SG], thr[SG]; bool indep = false;
#pragma unroll
                    for (int jj = 0; jj < SG; ++jj) {
                        const u32 ri = 63u - ((m[jj] >> 16) & 63u), li = m[jj] & 63u;
                        val[jj] = (int32_t)ri; thr[jj] = ri == li ? ANY : jj; indep = indep || ri == li;
                    }
                    int32_t T = 4096; u32 cbits = 0;      // (4096: unknown / nothing tracked, UNKI below)
#pragma unroll
                    for (int jj = 0; jj < SG; ++jj) { T = T < thr[jj] ? val[jj] : T; cbits |= T == jj + (int32_t)t - 1 ? 1u << jj : 0u; }
                    int32_t Tout = T;
                    auto shift_in = [&](int32_t to) -> int32_t { int32_t x = __shfl_up(to, 1, 64); if (lane == 0) x = carry; return x == 4096 ? 4096 : x - SG; };
                    int32_t Tin = shift_in(Tout);
                    if (__any((int)!indep)) {
                        for (;;) {
                            bool chg = false;
                            if (!indep) { int32_t tt = Tin;
#pragma unroll
                                for (int jj = 0; jj < SG; ++jj) tt = tt < thr[jj] ? val[jj] : tt;
                                chg = tt != Tout; Tout = tt; }
                            if (!__any((int)chg)) break;
                            Tin = shift_in(Tout);
                        }
                    }
                    carry = __builtin_amdgcn_readlane(Tout, 63);
                    if (rd >= 0) {
                        // my windows in front of the first one with a unique minimum started from "unknown": once more from what came in, as far as any lane needs it
                        bool alive = true; int32_t tt = Tin;
                        need_slow = need_slow || (Tin == 4096 && thr[0] != ANY && (own & 1u));
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) {
                            alive = alive && thr[jj] != ANY;
                            if (!__any((int)alive)) break;
                            tt = alive && tt < jj ? val[jj] : tt;
                            cbits |= alive && tt == jj + (int32_t)t - 1 ? 1u << jj : 0u;
                        }
                        put_cands(cbits & own, p0);
                    }
                    continue;
                }
                u32 h6[NH];                               // hash << 6 of the s-mer ending at p0 - (w - 1) + ii (garbage where no s-mer is: never inside a window that counts)
                if (use_lut) {
                    const u64 Y1 = W1 >> (16u - sm), Y0 = W0 >> (16u - sm);      // the s-mer ending at qh + ii = view indices [ii, ii + s): its last bit now lies at bit 48 - ii
#pragma unroll
                    for (int ii = 0; ii < NH; ++ii) {
                        const u32 ka = (u32)(Y1 >> (48 - ii)) & fm, kb = (u32)(Y0 >> (48 - ii)) & fm;
                        h6[ii] = (u32)lut[ka << sm | kb] << 6;
                    }
                } else {
                    u32 xs0 = 0, xs1 = 0;
                    for (u32 i = 0; i + 1 < sm; ++i) {        // the s - 1 bases in front of the first s-mer I need
                        const u32 c = (u32)((W1 >> (63u - i)) & 1ull) << 1 | (u32)((W0 >> (63u - i)) & 1ull);
                        xs0 = (xs0 << 2 | c) & smask; xs1 = xs1 >> 2 | (3u - c) << sshift;
                    }
                    const u64 V1 = W1 << (sm - 1), V0 = W0 << (sm - 1);      // position qh in bit 63: static bit indices from here on
#pragma unroll
                    for (int ii = 0; ii < NH; ++ii) {
                        const u32 c = (u32)((V1 >> (63 - ii)) & 1ull) << 1 | (u32)((V0 >> (63 - ii)) & 1ull);
                        xs0 = (xs0 << 2 | c) & smask; xs1 = xs1 >> 2 | (3u - c) << sshift;
                        h6[ii] = sync_hash32(xs0 < xs1 ? xs0 : xs1, smask) << 6;
                    }
                }
                // every window's rightmost and leftmost minimum (as END positions) and whether they are the same one: one unsigned min each over hash << 6 | index
                // (rightmost: 63 - index).  WMAX > SG: every window straddles the indices SG - 1 | SG — suffix minima up to SG - 1, prefix minima from SG, one min per window
                // val[jj]: what the tracked s-mer becomes at window jj when it changes, as an index of my hashes (the s-mer ending at qh + index): the leftmost minimum at a
                // read's first window, else the rightmost; K: the windows that FIX the state whatever came in (unique minimum, first window, no full l-mer: UNKI)
                constexpr int32_t UNKI = 4096;            // "unknown / nothing tracked" in index coordinates (no index compares equal or smaller)
                int32_t val[SG]; u32 K = ~vmask;
                {
                    u32 br[SG], bl[SG];
                    if constexpr (WMAX > 2) {
                        // blocks of w - 1 indices: a window (w indices) is the tail of one block and the head of the next, down to its last index: suffix minima inside
                        // every block, prefix minima inside every block, one min per window and side — 4 mins per hash instead of 2 (w - 1) per window
                        constexpr int B = WMAX - 1;
                        u32 sr[NH], sl[NH], pr[NH], pl[NH];
#pragma unroll
                        for (int i = NH - 1; i >= 0; --i) {
                            const u32 er = h6[i] | (u32)(63 - i), el = h6[i] | (u32)i;
                            const bool edge = i % B == B - 1 || i == NH - 1;
                            sr[i] = edge ? er : (er < sr[(i + 1) % NH] ? er : sr[(i + 1) % NH]); sl[i] = edge ? el : (el < sl[(i + 1) % NH] ? el : sl[(i + 1) % NH]);
                        }
#pragma unroll
                        for (int i = 0; i < NH; ++i) {
                            const u32 er = h6[i] | (u32)(63 - i), el = h6[i] | (u32)i;
                            const bool edge = i % B == 0;
                            pr[i] = edge ? er : (er < pr[(i + NH - 1) % NH] ? er : pr[(i + NH - 1) % NH]); pl[i] = edge ? el : (el < pl[(i + NH - 1) % NH] ? el : pl[(i + NH - 1) % NH]);
                        }
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) {      // window jj = indices [jj, jj + B]
                            const u32 a_ = sr[jj], b_ = pr[jj + B]; br[jj] = a_ < b_ ? a_ : b_;
                            const u32 c_ = sl[jj], d_ = pl[jj + B]; bl[jj] = c_ < d_ ? c_ : d_;
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) {
                            u32 x = h6[jj] | (u32)(63 - jj), y = h6[jj] | (u32)jj;
#pragma unroll
                            for (int d = 1; d < WMAX; ++d) { const u32 er = h6[jj + d] | (u32)(63 - jj - d), el = h6[jj + d] | (u32)(jj + d); x = er < x ? er : x; y = el < y ? el : y; }
                            br[jj] = x; bl[jj] = y;
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < SG; ++jj) {
                        const u32 ri = 63u - (br[jj] & 63u), li = bl[jj] & 63u;
                        const bool v = (vmask >> jj) & 1u, f = (fmask >> jj) & 1u;
                        K |= (ri == li || f) ? 1u << jj : 0u;
                        val[jj] = !v ? UNKI : (int32_t)(f ? li : ri);
                    }
                }
                // the steps from the state T in front of window `from` on (T: index of the tracked s-mer among my hashes, UNKI: unknown): window jj holds the indices
                // [jj, jj + w - 1] — the tracked one has left when T < jj —, and is a candidate when T is its index jj + t - 1.  cb / ub: bit jj = candidate / state unknown
                auto step = [&](int jj, int32_t& T, u32& cb, u32& ub) {
                    T = (((K >> jj) & 1u) || T < jj) ? val[jj] : T;
                    cb |= T == jj + (int32_t)t - 1 ? 1u << jj : 0u; ub |= T == UNKI ? 1u << jj : 0u;
                };
                int32_t T = UNKI; u32 cbits = 0, ubits = 0;
#pragma unroll
                for (int jj = 0; jj < SG; ++jj) step(jj, T, cbits, ubits);
                int32_t Tout = T;                          // (exact unless none of my windows fixes the state)
                const bool indep = (K & ((1u << SG) - 1u)) != 0u;
                auto shift_in = [&](int32_t to) -> int32_t {      // the state behind the lane below me (the round before, for lane 0), in MY index coordinates
                    int32_t x = __shfl_up(to, 1, 64); if (lane == 0) x = carry;
                    return x == UNKI ? UNKI : x - SG;
                };
                int32_t Tin = shift_in(Tout);
                if (__any((int)!indep)) {
                    // a lane without a fixing window (rare: eight windows with tied minima) passes on what comes in: until nothing changes any more
                    for (;;) {
                        bool chg = false;
                        if (!indep) { int32_t tt = Tin; u32 c2 = 0, u2 = 0;
#pragma unroll
                            for (int jj = 0; jj < SG; ++jj) step(jj, tt, c2, u2);
                            chg = tt != Tout; Tout = tt; }
                        if (!__any((int)chg)) break;
                        Tin = shift_in(Tout);
                    }
                }
                carry = __builtin_amdgcn_readlane(Tout, 63);
                if (rd >= 0) {
                    // my windows in front of the first fixing one took their state from UNKI: once more with what came in, as far as any lane of the wave needs it
                    const u32 k8 = K & ((1u << SG) - 1u);
                    const u32 dep = k8 ? (k8 & (0u - k8)) - 1u : (1u << SG) - 1u;      // my windows in front of the first fixing one (all of them valid: a window without a full l-mer fixes)
                    if (Tin != UNKI && dep) {
                        int32_t tt = Tin; u32 c2 = 0, u2 = 0;
#pragma unroll
                        for (int jj = 0; jj < SG; ++jj) { if (!__any((int)((dep >> jj) & 1u))) break; if ((dep >> jj) & 1u) step(jj, tt, c2, u2); }
                        cbits |= c2 & dep; ubits = (ubits & ~dep) | (u2 & dep);
                    }
                    need_slow = need_slow || (ubits & vmask & own) != 0u;
                    put_cands(cbits & vmask & own, p0);
                }
            }
            if (need_slow) S.misc[9] = 1;
        }
        tile_sync<NW>();
        if (S.misc[9]) { tile_sync<NW>(); run_slow_tile(); return; }
    }
    tile_sync<NW>();
    // the (unordered) candidate list: every thread expands the bitmap words 4 tid .. 4 tid + 3.  (Round 2 and the first version of this
    // round appended to the list inside the filter loop: one LDS fetch-add and a bit loop per step, 12 times per wave instead of once.)
    auto expand_list = [&]() {
        const uint4 cw = *(const uint4*)(S.c.cand + WPT * tid);
        const u32 w5[5] = {cw.x, cw.y, cw.z, cw.w, tid == TT - 1 ? S.c.cand[RW] : 0u};       // the last thread also takes word RW
        const u32 c = bs_popc(cw.x) + bs_popc(cw.y) + bs_popc(cw.z) + bs_popc(cw.w) + bs_popc(w5[4]);
        // one LDS fetch-add per WAVE: the lanes' counts are scanned with DPP and the last lane reserves the wave's slots (a fetch-add per thread with
        // candidates — about half of them, all on one address — was 0.3 - 1.7 % slower in three of three pairs, profiles/r04_l_micro_ab.txt)
        const u32 inc_c = wave_incl_scan(c);
        u32 wbase = 0;
        if (lane == 63) wbase = lds_fetch_add(&S.misc[17], inc_c);
        wbase = (u32)__builtin_amdgcn_readlane((int)wbase, 63);
        if (c) {
            u32 slot = wbase + inc_c - c;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                u32 w = w5[i];
                while (w) {
                    const u32 b = (u32)__clz(w); w &= ~(0x80000000u >> b);
                    if (slot < QCAP) S.c.list[slot] = (u16)(32 * (WPT * tid + i) + b - (BS_B - 1));
                    ++slot;
                }
            }
        }
    };
    expand_list();
    tile_sync<NW>();
    MDBG_STAMP(3);
    if (MDBG_HOT(a.stop_phase == 3, false)) { if (tid == 0) a.n_valid[gt] = 0; return; }

    // ---- phase 4: exact evaluation, ranks, records -------------------------------------------------------------------
    auto count_words = [&]() -> u32 {                                 // cpre[] <- exclusive counts of the bitmap per word; returns the total
        const uint4 cw = *(const uint4*)(S.c.cand + WPT * tid);
        const u32 c0 = bs_popc(cw.x), c1 = bs_popc(cw.y), c2 = bs_popc(cw.z), c3 = bs_popc(cw.w);
        const u32 last = tid == TT - 1 ? bs_popc(S.c.cand[RW]) : 0u;       // the last thread also takes word RW
        u32 total;
        const u32 o = tile_excl_scan<NW>(c0 + c1 + c2 + c3 + last, S.misc, total);
        const u32 o1 = o + c0, o2 = o1 + c1, o3 = o2 + c2;
        *(uint2*)(S.c.cpre + WPT * tid) = make_uint2(o | o1 << 16, o2 | o3 << 16);
        if (tid == TT - 1) S.c.cpre[RW] = (u16)(o3 + c3);
        tile_sync<NW>();
        return total;
    };
    // tile-relative raw position of dense position r.  The kept fraction is nearly uniform along a tile, so r * RW / H lands within a word
    // or two of the raw word that holds r (round 2 kept a table of first raw words per dense word: 2 KB of LDS and five instructions
    // per raw word in phase 2; that LDS is what admits a seventh workgroup per CU)
    const float raw_per_dense = (float)RW / (float)(H ? H : 1u);
    auto dense_to_raw = [&](u32 r) -> u32 {
        u32 w = (u32)((float)r * raw_per_dense);
        w = w < (u32)RW - 1u ? w : (u32)RW - 1u;
        while (S.rpre[w] > r) --w;                                      // rpre[0] = 0 <= r
        while (S.rpre[w + 1] <= r) ++w;                                 // rpre[RW] = H > r
        return 32 * w + bs_select_msb(S.kw[w], r - S.rpre[w]);
    };
    // exact 64-bit hash of the l-mer ending at dense position e: ntHash (src/read.rs:196), or — syncmers — the integer hash of the
    // canonical 2-bit l-mer (src/read.rs:300-318): the two code planes are interleaved through the 8 -> 16 bit spread table in t3
    auto exact = [&](u32 e) -> u64 {
        const u32 wi = e >> 5, s = e & 31;
        const u32* dw = S.dense + 2 * (DPAD + wi);
        const u32 v0 = bs_alignbit(dw[-2], dw[0], 31 - s), v1 = bs_alignbit(dw[-1], dw[1], 31 - s);      // bit u = dense position e - u
        if constexpr (SCHEME == 0) return bs_exact_hash<BS_GS, L>(v0, v1, S_t3);
        else {
            const u32 lm = Lr >= 32 ? 0xFFFFFFFFu : (1u << Lr) - 1u;
            const u32 r0 = (v0 ^ v1) & lm, r1 = v1 & lm;              // the reference's code bits of the base at distance u from the end
            const u16* sp = (const u16*)S_t3;
            const u32 lo = ((u32)sp[r0 & 255u] | (u32)sp[(r0 >> 8) & 255u] << 16) | ((u32)sp[r1 & 255u] | (u32)sp[(r1 >> 8) & 255u] << 16) << 1;
            const u32 hi = ((u32)sp[(r0 >> 16) & 255u] | (u32)sp[r0 >> 24] << 16) | ((u32)sp[(r1 >> 16) & 255u] | (u32)sp[r1 >> 24] << 16) << 1;
            const u64 lmask = (1ull << (2 * Lr)) - 1ull;
            const u64 xl0 = (u64)hi << 32 | lo;                        // base at distance u in bit pair u (the newest base lowest: read.rs xl[0])
            // xl[1]: complements (3 - code), pairs in reverse order inside the 2l-bit field
            const u64 cm = ~xl0 & lmask;
            u64 y = (u64)__brev((u32)cm) << 32 | (u64)__brev((u32)(cm >> 32));
            y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
            const u64 xl1 = y >> (64 - 2 * Lr);
            return sync_hash(xl0 < xl1 ? xl0 : xl1, lmask);
        }
    };
    // raw coordinates of a selected l-mer (src/read.rs:196-208): its FIRST base decides the read; an l-mer that runs over the next
    // read's start (a forced run start, so its dense rank is the number of kept bases in front of it) belongs to no read
    auto place = [&](u32 e, u64 h, CandOut& o) -> bool {
        const u32 sd = e - (Lr - 1);
        const int64_t rel_start = dense_to_raw(sd);
        u32 r; int64_t q0, q1;                                         // the read holding the first base; q0: its start, q1: the next read's, tile-relative
        if (MDBG_HOT(n_rs <= RS_CAP, true)) {
            u32 i = 0;
            while (i + 1 < n_rs && (int64_t)S.rs_rel[i + 1] <= rel_start) ++i;
            r = rl + i; q0 = i ? (int64_t)S.rs_rel[i] : S.rs0; q1 = i + 1 < n_rs ? (int64_t)S.rs_rel[i + 1] : (int64_t)RW * 32;
        } else {
            r = find_read(a.offsets, rl, rh_, (u64)(raw0 + rel_start)); q0 = (int64_t)a.offsets[r] - raw0;
            q1 = r < rh_ ? (int64_t)a.offsets[r + 1] - raw0 : (int64_t)RW * 32;
        }
        bool crosses = false;
        if (q1 < (int64_t)RW * 32) {
            const u32 w = (u32)q1 >> 5, b = (u32)q1 & 31;
            const u32 ds = S.rpre[w] + (b ? bs_popc(S.kw[w] >> (32 - b)) : 0u);
            crosses = ds <= e;
        }
        // (the outputs are written on every path: hipcc 7.2 zeroed a caller's variable that was assigned only after a successful return
        // for the lanes that came through the block above)
        o.hash = h; o.pos = (u32)(rel_start - q0); o.read = r + a.read_base;
        return !crosses;
    };
    auto clear_bit = [&](u32 e) { const u32 x = e + BS_B - 1; atomicAnd(&S.c.cand[x >> 5], ~(0x80000000u >> (x & 31))); };
    auto rank_of = [&](u32 e) -> u32 { const u32 x = e + BS_B - 1, D = x >> 5, b = x & 31; return S.c.cpre[D] + (b ? bs_popc(S.c.cand[D] >> (32 - b)) : 0u); };

    // One round over the candidates list[0 .. n): two stages per TT candidates — (1) one lane per candidate evaluates the exact hash
    // (about half fail: bit 55), the survivors are packed (wave-aggregated slot) into surv / their hashes; (2) one lane per SURVIVOR maps
    // it to raw coordinates and its read, so the longer half of the work runs on full waves.  Failures leave the bitmap; the popcount scan
    // of the bitmap then ranks the survivors in position order (every bit in front of a survivor of this round is final: earlier rounds
    // and this one) and they are written.  misc[18 .. 20] are zero on entry.  Returns the number of bits left in the bitmap.
    auto process_list = [&](u32 n) -> u32 {
        constexpr int NR = (QCAP + TT - 1) / TT;
        CandOut keep[NR] = {}; u32 keep_e[NR] = {}; u32 keep_ok = 0;
        u64* const s_h = (u64*)S.c.cpre;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if ((u32)(TT * i) < n) {
                if (i) tile_sync<NW>();                               // the previous sub-round's survivors have been read
                const u32 j = tid + TT * i;
                bool pass = false; u32 e = 0; u64 h = 0;
                if (j < n) { e = S.c.list[j]; h = exact(e); pass = h <= a.bound; if (!pass) clear_bit(e); }
                const u64 bal = __ballot(pass);
                u32 n_surv;
                if constexpr (NW == 1) {                               // one wave: its own ballot is the whole round
                    n_surv = (u32)__popcll(bal);
                    if (pass) { const u32 slot = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)); S.c.surv[slot] = (u16)e; s_h[slot] = h; }
                    tile_sync<NW>();
                } else {
                    if (bal) {
                        u32 base = 0;
                        if (lane == 0) base = atomicAdd(&S.misc[18 + i], (u32)__popcll(bal));
                        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                        if (pass) { const u32 slot = base + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)); S.c.surv[slot] = (u16)e; s_h[slot] = h; }
                    }
                    tile_sync<NW>();
                    n_surv = S.misc[18 + i];
                }
                if ((u32)tid < n_surv) {
                    e = S.c.surv[tid];
                    keep_e[i] = e;
                    if (place(e, s_h[tid], keep[i])) keep_ok |= 1u << i; else clear_bit(e);
                }
            }
        }
        tile_sync<NW>();
        MDBG_STAMP(4);
        if (MDBG_HOT(a.stop_phase == 4, false)) return 0u;            // (diagnostic: instruction counts of the exact evaluation + placement alone; uniform over the tile)
        const u32 left = count_words();
        MDBG_STAMP(5);
        if (MDBG_HOT(a.stop_phase == 5, false)) return 0u;
#pragma unroll
        for (int i = 0; i < NR; ++i) if ((keep_ok >> i) & 1u) {
            const u32 rank = rank_of(keep_e[i]);
            put_rec(a, slab, rank, keep[i].hash, keep[i].pos, keep[i].read);
            if (rank + 1 == left && a.last_read) a.last_read[gt] = keep[i].read;
        }
        return left;
    };

    const u32 n_cand = S.misc[17];
    if (MDBG_HOT(n_cand <= QCAP, true)) {
        // every candidate sits in the (unordered) list of phase 3
        const u32 nv = n_cand ? process_list(n_cand) : 0;
        if (tid == 0) { put_count(a, gt, nv, true); if (a.dbg) { a.dbg[(size_t)gt * 16 + 9] = n_cand; a.dbg[(size_t)gt * 16 + 10] = nv; } }
    } else {
        // Dense settings (more candidates than the list holds).  The candidates are dealt out evenly by their rank in the bitmap — a
        // thread finds the word of its first one by binary search over the per-word counts and walks on from there —, evaluated once
        // each, and written to the slab AT THEIR CANDIDATE RANK, the rejected ones with read = 0xFFFFFFFF: no lists, no rounds, no
        // barriers, no second evaluation; the gather squeezes the holes out (it is a copy anyway).  Round 2 validated all candidates in
        // one pass and evaluated the survivors again to write them; a first version of this round ran the two-stage rounds of the fast
        // path over stretches of the bitmap: 0.43 Tbases/s at d = 0.1, most of it spent at ~100 barriers per tile.
        tile_sync<NW>();
        if constexpr (SCHEME == 1) {
            // Syncmers: one position in w is a candidate (2,700 of a tile's 24,500 at l = 12 s = 4), and the density bound then keeps a few per cent of them.  Written
            // like the dense settings of the density scheme — every candidate a 16-byte slot of the slab, the rejected ones as holes — that was 43 KB per tile to HBM
            // and back through the gather: the kernel's longest phase at a quarter of its instructions.  So the candidates are first thinned out where they lie: every
            // thread takes its own words of the bitmap (no word is shared: plain stores), evaluates the exact hash of each candidate and drops the failures; what is left
            // mostly fits the list and goes the way of the sparse settings (survivors only, at their ranks).
            {
                const uint4 cw = *(const uint4*)(S.c.cand + WPT * tid);
                u32 w5[5] = {cw.x, cw.y, cw.z, cw.w, tid == TT - 1 ? S.c.cand[RW] : 0u};
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    u32 w = w5[i], keep = w;
                    while (w) {
                        const u32 b = (u32)__clz(w), bit = 0x80000000u >> b; w &= ~bit;
                        if (exact((u32)(32 * (WPT * tid + i)) + b - (BS_B - 1)) > a.bound) keep &= ~bit;
                    }
                    w5[i] = keep;
                }
                *(uint4*)(S.c.cand + WPT * tid) = make_uint4(w5[0], w5[1], w5[2], w5[3]);
                if (tid == TT - 1) S.c.cand[RW] = w5[4];
                if (tid == 0) S.misc[17] = 0;
            }
            tile_sync<NW>();
            expand_list();
            tile_sync<NW>();
            const u32 n2 = S.misc[17];
            if (n2 <= QCAP) {
                const u32 nv = n2 ? process_list(n2) : 0;
                if (tid == 0) { put_count(a, gt, nv, true); if (a.dbg) { a.dbg[(size_t)gt * 16 + 9] = n_cand; a.dbg[(size_t)gt * 16 + 10] = nv; } }
                return;
            }
            tile_sync<NW>();
        }
        const u32 C = count_words();
        const u32 per = (C + TT - 1) / TT;
        const u32 r0 = (u32)tid * per < C ? (u32)tid * per : C, r1 = r0 + per < C ? r0 + per : C;
        u32 surv = 0;
        if (r0 < r1) {
            u32 lo = 0, hi = RW;                                       // largest D with cpre[D] <= r0: the word that holds candidate r0
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (S.c.cpre[mid] <= r0) lo = mid; else hi = mid - 1; }
            u32 D = lo, w = S.c.cand[D];
            for (u32 skip = r0 - S.c.cpre[D]; skip; --skip) w &= ~(0x80000000u >> (u32)__clz(w));
            for (u32 r = r0; r < r1; ++r) {
                while (!w) { ++D; w = S.c.cand[D]; }
                const u32 b = (u32)__clz(w); w &= ~(0x80000000u >> b);
                const u32 e = 32 * D + b - (BS_B - 1);
                const u64 h = exact(e);
                CandOut o{}; bool ok = false;
                if (h <= a.bound) ok = place(e, h, o);
                Rec rec; rec.hash = ok ? o.hash : 0ull; rec.pos = ok ? o.pos : 0u; rec.read = ok ? o.read : 0xFFFFFFFFu;
                surv += ok ? 1u : 0u;
                if (r < a.slab_cap) slab[r] = rec;
            }
        }
        u32 nv;
        tile_excl_scan<NW>(surv, S.misc, nv);
        if (tid == 0) {
            a.n_valid[gt] = nv; if (a.n_scan) a.n_scan[gt] = C;
            if (a.block_sum && nv) atomicAdd(a.block_sum + ((gt - a.tile0) >> SCAN_GRAN_LOG), (unsigned long long)nv);
            if (a.last_read) a.last_read[gt] = nv ? LAST_IN_SLAB : LAST_NONE;
            if (C > a.slab_cap) atomicMax(a.over_max, C);
            if (a.dbg) { a.dbg[(size_t)gt * 16 + 9] = n_cand; a.dbg[(size_t)gt * 16 + 10] = nv; }
        }
    }
    MDBG_STAMP(7);
#undef MDBG_STAMP
}

// ---- gather: per-tile records -> final position-ordered arrays ---------------------------------------------
// exclusive scan of n_valid over the tiles (carry[0] in/out = running total), three small kernels:
// sums of 1024-tile blocks, scan of the block sums by one workgroup, per-block scan + base.
__global__ __launch_bounds__(256) void tile_scan_sums_kernel(u32 n, const u32* __restrict__ n_valid, u64* __restrict__ block_sum) {
    __shared__ u32 ws[4];
    const u32 i0 = blockIdx.x * 1024 + threadIdx.x * 4;
    u32 v = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i0 + q < n) v += n_valid[i0 + q];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (u64)ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void tile_scan_top_kernel(u32 n_blocks, u64* __restrict__ block_sum, u64* __restrict__ carry) {
    __shared__ u64 ws[4]; __shared__ u64 run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) run = carry[0];
    __syncthreads();
    for (u32 i0 = 0; i0 < n_blocks; i0 += 256) {
        const u32 i = i0 + tid;
        const u64 v = i < n_blocks ? block_sum[i] : 0;
        u64 inc = v;
        for (int d = 1; d < 64; d <<= 1) { const u64 o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        u64 b = run, tot = 0;
        for (int q = 0; q < 4; ++q) { if (q < wv) b += ws[q]; tot += ws[q]; }
        if (i < n_blocks) block_sum[i] = b + inc - v;
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) carry[0] = run;
}
// self_base: block_base holds the plain block sums (at most 1024 blocks of 1024 tiles, no tile_scan_top launch): the workgroup adds up the
// sums in front of its block itself (the gather's last wave then moves the running total on)
// self_base == 2: block_base holds the sums of SCAN_GRAN-tile granules (the tiles added their counts up themselves, SketchArgs::block_sum): 1024 / SCAN_GRAN of them per workgroup
__global__ __launch_bounds__(256) void tile_scan_final_kernel(u32 n, const u32* __restrict__ n_valid, const u64* __restrict__ block_base, u64* __restrict__ tile_base,
                                                              u32 self_base, const u64* __restrict__ carry) {
    __shared__ u32 tmp[8];
    __shared__ u64 ws[4];
    const u32 i0 = blockIdx.x * 1024 + threadIdx.x * 4;
    u32 v[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = i0 + q < n ? n_valid[i0 + q] : 0u; mine += v[q]; }
    u64 bb;
    if (self_base) {
        u64 x = 0;
        if (self_base == 2) { const u32 ng = blockIdx.x * (1024 / SCAN_GRAN); for (u32 g = threadIdx.x; g < ng; g += 256) x += block_base[g]; }
        else {
#pragma unroll
            for (u32 q = 0; q < 4; ++q) if (4 * threadIdx.x + q < blockIdx.x) x += block_base[4 * threadIdx.x + q];
        }
        for (int d = 32; d; d >>= 1) x += __shfl_down(x, d, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = x;
        __syncthreads();
        bb = carry[0] + ws[0] + ws[1] + ws[2] + ws[3];
    } else bb = block_base[blockIdx.x];
    u32 total;
    u64 b = bb + block_excl_scan_256(mine, tmp, total);

#pragma unroll
    for (int q = 0; q < 4; ++q) { if (i0 + q < n) tile_base[i0 + q] = b; b += v[q]; }
}
// one wave per tile of the launch (tiles [tile0, tile0 + n)); a tile whose records did not fit is skipped (the host retries).
// The per-read offsets come out of the same pass (round 2 ran a second kernel over the gathered read indices): off[slot] = first
// index i with mread[i] >= slot for the batch's slots [slot0, slot0 + n_reads].  A record whose read differs from the record in front of
// it writes off[] for its read and for the empty reads in between; "the record in front" of a tile's first record is the last record of
// the nearest non-empty tile in front (a walk over n_valid, amortised one step per tile), or — in front of the launch — the last
// entry an earlier launch of the batch wrote; the wave of the batch's LAST tile also fills the entries behind the last record.
struct GatherArgs {
    u32 tile0, n; const Rec* slab; u32 slab_cap; const u32* n_valid; const u32* n_scan; const u64* tile_base;      // n_scan: see SketchArgs (null: every slab is compact)
    const u32* last_read;                                         // see SketchArgs (null: always look at the slabs)
    u64* out_hash; u32* out_pos; u32* out_read; u64 out_cap;
    u64 m0; u32 slot0, n_reads; u64* off; u32 last_launch;        // m0: first store index of the batch
    u64* carry;                                                   // non-null: <- running total behind this launch's last tile (the scan ran without tile_scan_top)
    u32 tiles_per_wave;                                           // gather_multi_kernel: consecutive tiles one wave takes (small tiles)
};
constexpr u32 REC_REJECTED = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs g) {
    const u32 b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= g.n) return;
    const u32 lane = threadIdx.x & 63;
    // everything that depends on b alone is requested at once (the kernel is a chain of dependent round trips, 13 waves deep per SIMD):
    // the counts, the base, the read in front, and the first 128 slots of the slab whatever they hold
    const Rec* s = g.slab + (size_t)b * g.slab_cap;
    const u32 nv = g.n_valid[g.tile0 + b];
    const u32 ns = g.n_scan ? g.n_scan[g.tile0 + b] : 0u;
    const u64 base = g.tile_base[b];
    u32 lr = LAST_IN_SLAB;
    if (g.last_read && b) lr = g.last_read[g.tile0 + b - 1];
    Rec pre[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { pre[u] = Rec{}; if (64 * u + lane < g.slab_cap) pre[u] = s[64 * u + lane]; }
    const u32 slots = ns ? ns : nv;                                   // slab slots to look at (ns != 0: some hold rejected candidates)
    const bool tail = g.last_launch && b == g.n - 1;
    if (g.carry && b == g.n - 1 && lane == 0) *g.carry = base + nv;
    if ((!nv && !tail) || slots > g.slab_cap) return;
    int64_t prev = (int64_t)g.slot0 - 1;                              // read of the record in front of this tile's first
    if (b && lr < LAST_IN_SLAB) prev = (int64_t)lr;
    else {
        int64_t q = (int64_t)b - 1;
        while (q >= 0 && g.n_valid[g.tile0 + q] == 0) --q;
        if (q >= 0) {
            const u32 lq = g.last_read ? g.last_read[g.tile0 + q] : LAST_IN_SLAB;
            if (lq < LAST_IN_SLAB) prev = (int64_t)lq;
            else {
                const u32 cq = g.n_valid[g.tile0 + q], sq = g.n_scan ? g.n_scan[g.tile0 + q] : 0u;
                int64_t j = (int64_t)(sq ? sq : cq) - 1;
                if (j < (int64_t)g.slab_cap) {
                    const Rec* t = g.slab + (size_t)q * g.slab_cap;
                    while (j >= 0 && t[j].read == REC_REJECTED) --j;
                    if (j >= 0) prev = (int64_t)t[j].read;
                }
            }
        } else { const u64 b0 = g.tile_base[0]; if (b0 > g.m0 && b0 <= g.out_cap) prev = (int64_t)g.out_read[b0 - 1]; }
    }
    const int64_t lo = (int64_t)g.slot0 - 1, hi = (int64_t)g.slot0 + g.n_reads;
    if (prev < lo) prev = lo;
    u64 out = base;
    for (u32 j0 = 0; j0 < slots; j0 += 64) {
        const u32 j = j0 + lane;
        Rec r{}; r.read = REC_REJECTED;
        if (j < slots) r = pre[0];
        // two rounds ahead stay in flight (dense settings: thousands of records per tile, one dependent round trip per 64 of them otherwise)
        pre[0] = pre[1];
        if (j + 128 < slots && j + 128 < g.slab_cap) pre[1] = s[j + 128];
        const bool valid = r.read != REC_REJECTED;
        const u64 bal = __ballot(valid);
        const u64 lower = bal & ((1ull << lane) - 1ull);
        const u32 from_lane = __shfl(r.read, lower ? 63 - __clzll((long long)lower) : 0, 64);      // the valid record in front of mine in this round
        const int64_t pr = lower ? (int64_t)from_lane : prev;
        const u32 last_valid = __shfl(r.read, bal ? 63 - __clzll((long long)bal) : 0, 64);
        if (valid) {
            const u64 idx = out + (u64)__popcll(lower);
            if (idx < g.out_cap) { g.out_hash[idx] = r.hash; g.out_pos[idx] = r.pos; g.out_read[idx] = r.read; }
            int64_t cur = (int64_t)r.read; if (cur > hi) cur = hi;
            for (int64_t x = pr + 1; x <= cur; ++x) g.off[x] = idx;
        }
        if (bal) prev = (int64_t)last_valid;
        out += (u64)__popcll(bal);
    }
    if (tail) {
        int64_t last = prev < lo ? lo : prev;
        for (int64_t x = last + 1 + lane; x <= hi; x += 64) g.off[x] = out;
    }
}

// Small tiles (a wave tile holds a few dozen records at the usual densities): one wave takes T consecutive tiles and walks their slab
// slots as ONE flattened sequence, so the stores stay 64 records wide and the chain of dependent round trips (counts -> records ->
// stores) is paid once per T tiles.  Same contract as gather_kernel.
__global__ __launch_bounds__(256) void gather_multi_kernel(GatherArgs g) {
    const u32 T0 = g.tiles_per_wave;
    const u32 b0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * T0;       // first tile of the wave (launch-local)
    if (b0 >= g.n) return;
    const u32 lane = threadIdx.x & 63;
    const u32 T = g.n - b0 < T0 ? g.n - b0 : T0;                      // (<= 64)
    u32 nv = 0, ns = 0;
    if (lane < T) { nv = g.n_valid[g.tile0 + b0 + lane]; ns = g.n_scan ? g.n_scan[g.tile0 + b0 + lane] : 0u; }
    const u64 base = g.tile_base[b0];
    u32 lr = LAST_IN_SLAB;
    if (g.last_read && b0) lr = g.last_read[g.tile0 + b0 - 1];
    u32 slots = ns ? ns : nv;                                         // slab slots to look at (ns != 0: some hold rejected candidates)
    // a tile whose records did not fit its slab is skipped, and the later tiles of the same wave then write at output indices that are short by its records: the store of a
    // launch with an overflowing tile is NOT valid.  That is the contract, not an accident: the tile kernel of the same launch has raised over_max for that tile, and
    // sketch_device_impl reads it before it commits anything (attempt loop: larger slabs, the whole batch again); nothing reads the store in between.  gather_kernel
    // skips such a tile the same way (tests/test_gpu_round5.py forces the overflow on both).
    if (slots > g.slab_cap) slots = 0;
    const u32 inc = wave_incl_scan(slots);                            // inclusive prefix over the wave's tiles (lanes >= T: the total)
    const u32 S = (u32)__builtin_amdgcn_readlane((int)inc, 63);
    const u32 nv_all = (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(nv), 63);
    const bool tail = g.last_launch && b0 + T == g.n;
    if (g.carry && b0 + T == g.n && lane == 0) *g.carry = base + nv_all;
    if (!S && !tail) return;
    const Rec* const s0 = g.slab + (size_t)b0 * g.slab_cap;
    // slot j of the flattened sequence -> its record (tile = number of prefixes <= j)
    auto fetch = [&](u32 j) -> Rec {
        Rec r{}; r.read = REC_REJECTED;
        u32 ti = 0, ex = 0;                                           // (found by every lane: the loop and its lane reads are wave-uniform)
        for (u32 t = 0; t + 1 < T; ++t) { const u32 it = (u32)__builtin_amdgcn_readlane((int)inc, (int)t); if (j >= it) { ti = t + 1; ex = it; } }
        if (j < S) r = s0[(size_t)ti * g.slab_cap + (j - ex)];
        return r;
    };
    Rec cur = fetch(lane);
    int64_t prev = (int64_t)g.slot0 - 1;                              // read of the record in front of this wave's first
    if (b0 && lr < LAST_IN_SLAB) prev = (int64_t)lr;
    else {
        int64_t q = (int64_t)b0 - 1;
        while (q >= 0 && g.n_valid[g.tile0 + q] == 0) --q;
        if (q >= 0) {
            const u32 lq = g.last_read ? g.last_read[g.tile0 + q] : LAST_IN_SLAB;
            if (lq < LAST_IN_SLAB) prev = (int64_t)lq;
            else {
                const u32 cq = g.n_valid[g.tile0 + q], sq = g.n_scan ? g.n_scan[g.tile0 + q] : 0u;
                int64_t j = (int64_t)(sq ? sq : cq) - 1;
                if (j < (int64_t)g.slab_cap) {
                    const Rec* t = g.slab + (size_t)q * g.slab_cap;
                    while (j >= 0 && t[j].read == REC_REJECTED) --j;
                    if (j >= 0) prev = (int64_t)t[j].read;
                }
            }
        } else { const u64 bb = g.tile_base[0]; if (bb > g.m0 && bb <= g.out_cap) prev = (int64_t)g.out_read[bb - 1]; }
    }
    const int64_t lo = (int64_t)g.slot0 - 1, hi = (int64_t)g.slot0 + g.n_reads;
    if (prev < lo) prev = lo;
    u64 out = base;
    for (u32 j0 = 0; j0 < S; j0 += 64) {
        const Rec r = cur;
        if (j0 + 64 < S) cur = fetch(j0 + 64 + lane);                 // the next round is in flight while this one is placed
        const bool valid = r.read != REC_REJECTED;
        const u64 bal = __ballot(valid);
        const u64 lower = bal & ((1ull << lane) - 1ull);
        const u32 from_lane = __shfl(r.read, lower ? 63 - __clzll((long long)lower) : 0, 64);      // the valid record in front of mine in this round
        const int64_t pr = lower ? (int64_t)from_lane : prev;
        const u32 last_valid = __shfl(r.read, bal ? 63 - __clzll((long long)bal) : 0, 64);
        if (valid) {
            const u64 idx = out + (u64)__popcll(lower);
            if (idx < g.out_cap) { g.out_hash[idx] = r.hash; g.out_pos[idx] = r.pos; g.out_read[idx] = r.read; }
            int64_t c = (int64_t)r.read; if (c > hi) c = hi;
            for (int64_t x = pr + 1; x <= c; ++x) g.off[x] = idx;
        }
        if (bal) prev = (int64_t)last_valid;
        out += (u64)__popcll(bal);
    }
    if (tail) {
        int64_t last = prev < lo ? lo : prev;
        for (int64_t x = last + 1 + lane; x <= hi; x += 64) g.off[x] = out;
    }
}

// Error path only (a byte outside ACGTN was seen somewhere in the batch): the reference's exact rule — nthash panics iff a
// read whose HPC string has at least l bases holds such a byte (src/read.rs:157-174 decides the length).  One wave per
// read; *which = smallest offending read index (~0: none).
template <class Src>
__global__ __launch_bounds__(256) void alphabet_rule_kernel(Src src, const u64* __restrict__ off, u32 n_reads, u32 l, u32 hpc,
                                                            unsigned long long* __restrict__ which) {
    const u32 r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const u32 lane = threadIdx.x & 63;
    const u64 a = off[r], b = off[r + 1];
    u64 kept = 0; bool bad = false;
    for (u64 p = a + lane; p < b; p += 64) {
        const u8 c = src.at(p);
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
        if (!hpc || p == a || !(c == src.at(p - 1) && in_hpc_set(c))) ++kept;
    }
    for (int d = 32; d; d >>= 1) kept += __shfl_down(kept, d, 64);
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0 && any_bad && kept >= l) atomicMin(which, (unsigned long long)r);
}

// FMT_PLANES: marks the tiles whose staged range [t * STRIDE - HALO, (t + 1) * STRIDE) holds a listed exception
__global__ void tile_flags_kernel(const u64* __restrict__ exc_pos, u32 n_exc, u32 n_tiles, u8* __restrict__ flags, u64 stride, u64 halo) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_exc) return;
    const u64 p = exc_pos[i];
    const u64 t = p / stride;
    if (t < n_tiles) flags[t] = 1;
    if (t + 1 < n_tiles && p + halo >= (t + 1) * stride) flags[t + 1] = 1;      // falls into the next tile's look-back window
}

// ---- ASCII -> 2-bit planes on the device (mdbg_pack_device) -------------------------------------------------------
// one thread per 32 bases; bytes outside ACGT are appended (unordered) to the exception list
__global__ __launch_bounds__(256) void pack_planes_kernel(const u8* __restrict__ bases, u64 n_bases, uint2* __restrict__ words,
                                                          u64* __restrict__ exc_pos, u8* __restrict__ exc_val, u64 exc_cap,
                                                          unsigned long long* __restrict__ n_exc) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 p0 = w * 32;
    if (p0 >= n_bases) return;
    u32 bad = 0, hp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const u64 p = p0 + 16 * h;
        u32 w4[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        if (p + 16 <= n_bases) { const uint4 q = *(const uint4*)(bases + p); w4[0] = q.x; w4[1] = q.y; w4[2] = q.z; w4[3] = q.w; }
        else for (int i = 0; i < 16; ++i) if (p + i < n_bases) w4[i >> 2] = (w4[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((u32)bases[p + i] << (8 * (i & 3)));
        hp[h] = ascii16_to_hp(make_uint4(w4[0], w4[1], w4[2], w4[3]), bad);
    }
    // half planes are MSB first; the external layout has base i in bit i
    words[w] = make_uint2(__brev((hp[0] & 0xFFFF0000u) | (hp[1] >> 16)), __brev((hp[0] << 16) | (hp[1] & 0xFFFFu)));
    if (bad) {
        for (int i = 0; i < 32 && p0 + i < n_bases; ++i) {
            const u8 c = bases[p0 + i];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
                const u64 slot = atomicAdd(n_exc, 1ull);
                if (slot < exc_cap) { exc_pos[slot] = p0 + i; exc_val[slot] = c; }
            }
        }
    }
}
void launch_pack_planes(const u8* bases, u64 n_bases, uint2* words, u64* exc_pos, u8* exc_val, u64 exc_cap, unsigned long long* n_exc, hipStream_t s) {
    const u64 nw = (n_bases + 31) / 32;
    if (nw) hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, bases, n_bases, words, exc_pos, exc_val, exc_cap, n_exc);
}

// ---- host launchers -------------------------------------------------------------------------------
void launch_bread(const u64* offsets, u32 n_reads, u64 n_bases, u32 n_tiles, u32* bread, TileRec* recs, const SketchInit& init, const TileShape& sh, hipStream_t s) {
    const u32 n = n_tiles + 2;
    hipLaunchKernelGGL(tile_rec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, offsets, n_reads, n_bases, n_tiles, bread, recs, init, (int64_t)sh.stride, (int64_t)sh.halo);
}
void launch_tile_flags(const u64* exc_pos, u32 n_exc, u32 n_tiles, u8* flags, const TileShape& sh, hipStream_t s) {
    if (n_exc) hipLaunchKernelGGL(tile_flags_kernel, dim3((n_exc + 255) / 256), dim3(256), 0, s, exc_pos, n_exc, n_tiles, flags, (u64)sh.stride, (u64)sh.halo);
}
// The tile geometry of a context.  256-lane tiles are the default: the wave tiles (MDBG_TILE = "1x4": four to a workgroup, "1x1": one
// 64-lane workgroup each) were built to take the ~20 workgroup barriers out of a tile's life, and measured SLOWER by 7 - 9 % on configs[2]
// (profiles/r04_b_tile_shapes_ab.txt): the kernel follows its VALU instruction count, and a wave tile spends more of them per base (four
// filter steps of 63 words for 193 dense words, a placement pass per 24 instead of 97 survivors, twice the look-back share).
// The switch is read once; the syncmer scheme only exists on the 256-lane tiles.
// Since round 5 the wave-tile kernels are only compiled with -DMDBG_WAVE_TILES (scratch/build_variant.sh wave_tiles -DMDBG_WAVE_TILES; mdbg_build_flags() bit 0):
// the default library carries ONE tile shape, MDBG_TILE is ignored by it.
TileShape tile_shape_for(u32 scheme) {
    u32 nw = 4, tpw = 1;
#ifdef MDBG_WAVE_TILES
    static const char* const env = getenv("MDBG_TILE");
    if (env && env[0] == '1' && env[1] == 'x' && env[2] == '4') { nw = 1; tpw = 4; }
    else if (env && env[0] == '1' && env[1] == 'x' && env[2] == '1') { nw = 1; tpw = 1; }
#endif
    if (scheme == 1) { nw = 4; tpw = 1; }
    TileShape sh; sh.nw = nw; sh.tpw = tpw;
    sh.stride = nw == 4 ? (u32)TileGeo<4>::STRIDE : (u32)TileGeo<1>::STRIDE; sh.halo = nw == 4 ? (u32)TileGeo<4>::HALO_BASES : (u32)TileGeo<1>::HALO_BASES;
    return sh;
}
void launch_alphabet_rule(const SketchArgs& a, unsigned long long* which, hipStream_t s) {
    if (!a.n_reads) return;
    const dim3 g((a.n_reads + 3) / 4), b(256);
    if (a.fmt == FMT_ASCII) hipLaunchKernelGGL(alphabet_rule_kernel<AsciiSrc>, g, b, 0, s, AsciiSrc{a.bases}, a.offsets, a.n_reads, a.l, a.hpc, which);
    else hipLaunchKernelGGL(alphabet_rule_kernel<PlaneSrc>, g, b, 0, s, PlaneSrc{a.planes, a.exc_pos, a.exc_val, a.n_exc}, a.offsets, a.n_reads, a.l, a.hpc, which);
}

template <int L> static void launch_bs(const SketchArgs& a, u32 n_tiles, const TileShape& sh, hipStream_t s) {
#ifdef MDBG_WAVE_TILES
    if (sh.nw == 1 && sh.tpw == 1) { hipLaunchKernelGGL((sketch_bs_kernel<L, 0, 1, 1, 1>), dim3(n_tiles), dim3(64), 0, s, a); return; }
    if (sh.nw == 1) { hipLaunchKernelGGL((sketch_bs_kernel<L, 0, 1, 1, 4>), dim3((n_tiles + 3) / 4), dim3(256), 0, s, a); return; }
#endif
    (void)sh;
    hipLaunchKernelGGL((sketch_bs_kernel<L, 0, 1, 4, 1>), dim3(n_tiles), dim3(256), 0, s, a);
}
// one launch covers the whole batch (launch boundaries would only re-synchronise the workgroups' phases); n_tiles = tiles to run from a.tile0
void launch_sketch(SketchArgs a, u32 n_tiles, const TileShape& sh, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (!n_tiles) return;
    a.tile_end = a.tile0 + n_tiles;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    if (a.scheme == 1) {                              // syncmers: l is a run-time value, the window a compile-time one
#ifndef MDBG_ONLY_L
        switch (a.l - a.s + 1) {
#define MDBG_W(n) case n: hipLaunchKernelGGL((sketch_bs_kernel<0, 1, n>), dim3(n_tiles), dim3(256), 0, s, a); break;
            MDBG_W(1) MDBG_W(2) MDBG_W(3) MDBG_W(4) MDBG_W(5) MDBG_W(6) MDBG_W(7) MDBG_W(8) MDBG_W(9) MDBG_W(10) MDBG_W(11) MDBG_W(12) MDBG_W(13) MDBG_W(14) MDBG_W(15) MDBG_W(16)
            MDBG_W(17) MDBG_W(18) MDBG_W(19) MDBG_W(20) MDBG_W(21) MDBG_W(22) MDBG_W(23) MDBG_W(24) MDBG_W(25) MDBG_W(26) MDBG_W(27) MDBG_W(28) MDBG_W(29) MDBG_W(30) MDBG_W(31)
#undef MDBG_W
            default: hipLaunchKernelGGL((sketch_bs_kernel<0, 1, 32>), dim3(n_tiles), dim3(256), 0, s, a); break;
        }
#endif
    }
    else switch (a.l) {
#define MDBG_L(n) case n: launch_bs<n>(a, n_tiles, sh, s); break;
#ifdef MDBG_ONLY_L                                    // quick experiment builds: one l, no syncmers
        MDBG_L(MDBG_ONLY_L)
        default: break;
#else
        MDBG_L(2) MDBG_L(3) MDBG_L(4) MDBG_L(5) MDBG_L(6) MDBG_L(7) MDBG_L(8) MDBG_L(9) MDBG_L(10) MDBG_L(11) MDBG_L(12) MDBG_L(13)
        MDBG_L(14) MDBG_L(15) MDBG_L(16) MDBG_L(17) MDBG_L(18) MDBG_L(19) MDBG_L(20) MDBG_L(21) MDBG_L(22) MDBG_L(23) MDBG_L(24)
        MDBG_L(25) MDBG_L(26) MDBG_L(27) MDBG_L(28) MDBG_L(29) MDBG_L(30) MDBG_L(31) MDBG_L(32)
        default: launch_bs<32>(a, n_tiles, sh, s); break;      // l > 32: every tile takes the generic walker (force_slow is set by the host)
#endif
#undef MDBG_L
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
}

// scan + gather of the tiles [tile0, tile0 + n) of one launch; carry[0] in/out = running total of minimizers
// out[i] = carry[0] + sum of v[0 .. i) (n values; scan_tmp: n / 1024 + 2 u64); carry[0] <- carry[0] + total
void launch_excl_scan_u32(const u32* v, u32 n, u64* scan_tmp, u64* out, u64* carry, hipStream_t s) {
    if (!n) return;
    const u32 nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(tile_scan_sums_kernel, dim3(nb), dim3(256), 0, s, n, v, scan_tmp);
    hipLaunchKernelGGL(tile_scan_top_kernel, dim3(1), dim3(256), 0, s, nb, scan_tmp, carry);
    hipLaunchKernelGGL(tile_scan_final_kernel, dim3(nb), dim3(256), 0, s, n, v, scan_tmp, out, 0u, carry);
}
// gran_sum (non-null): the sums of the launch's SCAN_GRAN-tile granules (the tiles added their counts up themselves, SketchArgs::block_sum): the scan is ONE launch then
void launch_gather(const GatherArgs& g, u64* scan_tmp, u64* tile_base, u64* carry, hipStream_t s, const u64* gran_sum = nullptr) {
    if (!g.n) return;
    const u32 n = g.n, nb = (n + 1023) / 1024;
    const u32 self_base = nb <= 1024 ? 1u : 0u;     // up to 1,048,576 tiles (34 Gbases) per launch: no scan of the block sums by a kernel of its own
    if (gran_sum && self_base) hipLaunchKernelGGL(tile_scan_final_kernel, dim3(nb), dim3(256), 0, s, n, g.n_valid + g.tile0, gran_sum, tile_base, 2u, carry);
    else {
        hipLaunchKernelGGL(tile_scan_sums_kernel, dim3(nb), dim3(256), 0, s, n, g.n_valid + g.tile0, scan_tmp);
        if (!self_base) hipLaunchKernelGGL(tile_scan_top_kernel, dim3(1), dim3(256), 0, s, nb, scan_tmp, carry);
        hipLaunchKernelGGL(tile_scan_final_kernel, dim3(nb), dim3(256), 0, s, n, g.n_valid + g.tile0, scan_tmp, tile_base, self_base, carry);
    }
    GatherArgs a = g; a.tile_base = tile_base; a.carry = self_base ? carry : nullptr;
    if (a.tiles_per_wave > 1) { const u32 nw = (n + a.tiles_per_wave - 1) / a.tiles_per_wave; hipLaunchKernelGGL(gather_multi_kernel, dim3((nw + 3) / 4), dim3(256), 0, s, a); }
    else hipLaunchKernelGGL(gather_kernel, dim3((n + 3) / 4), dim3(256), 0, s, a);
}

// ---- --lmer-counts (src/read.rs:200-205, src/minimizers.rs:53-113): keep a selected minimizer only if its l-mer is in a given set ---------
// The set is the reference's minimizer_to_int restricted to l-mers over ACGT (a read's l-mer with any other byte is in no counts file of a
// k-mer counter), both orientations listed, as 2-bit codes: first base in the highest of the 2l bits, base code (ascii >> 1) & 3.  It lives
// in an open-addressing table (LMERSET_EMPTY = free; the all-ones code, "G" x 32, is kept apart).  The filter runs on the (few) minimizers
// that passed the density threshold: it re-reads the l-mer of each from the batch's bases (its HPC walk forward from the recorded raw
// position), looks the code up, and compacts the survivors through the tile slabs + gather that the sketch uses.
constexpr u64 LMERSET_EMPTY = ~0ull;
struct LmerFilterArgs {
    const u64* set; u64 set_mask; u32 has_all_ones;
    const u64* mh; const u32* mpos; const u32* mread; u64 m0, m1;      // the batch's minimizers in the resident store
    const u64* offsets; u32 slot0; u32 l; u32 hpc;
    Rec* slab; u32* n_valid;                                           // [n_blocks][256], [n_blocks]
};
__host__ __device__ inline u64 lmerset_home(u64 code, u64 mask) { return fmix64(code) & mask; }
template <bool HPC, class Src>
__device__ inline bool lmer_code_at(const Src& s, u64 rlo, u64 rhi, u64 p, u32 l, u64& code) {
    u64 q = p, c64 = 0;
    for (u32 j = 0; j < l; ++j) {
        if (q >= rhi) return false;
        const u8 c = s.at(q);
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
        c64 = (c64 << 2) | (u64)((c >> 1) & 3);
        ++q;
        if (HPC) while (q < rhi && s.at(q) == c) ++q;                  // c is in the HPC set: the rest of its run is dropped (read.rs:163-167)
    }
    code = c64;
    return true;
}
template <bool HPC, class Src>
__global__ __launch_bounds__(256) void lmer_filter_kernel(LmerFilterArgs a, Src src) {
    __shared__ u32 tmp[8];
    const u64 i = a.m0 + (u64)blockIdx.x * 256 + threadIdx.x;
    u32 keep = 0; Rec rec{};
    if (i < a.m1) {
        rec.hash = a.mh[i]; rec.pos = a.mpos[i]; rec.read = a.mread[i];
        const u32 r = rec.read - a.slot0;
        const u64 rlo = a.offsets[r], rhi = a.offsets[r + 1];
        u64 code;
        if (lmer_code_at<HPC>(src, rlo, rhi, rlo + rec.pos, a.l, code)) {
            if (code == LMERSET_EMPTY) keep = a.has_all_ones;
            else for (u64 h = lmerset_home(code, a.set_mask);; h = (h + 1) & a.set_mask) {
                const u64 v = a.set[h];
                if (v == code) { keep = 1; break; }
                if (v == LMERSET_EMPTY) break;
            }
        }
    }
    u32 total;
    const u32 rank = block_excl_scan_256(keep, tmp, total);
    if (keep) a.slab[(size_t)blockIdx.x * 256 + rank] = rec;
    if (threadIdx.x == 0) a.n_valid[blockIdx.x] = total;
}
u32 lmer_filter_blocks(u64 n_minimizers) { return (u32)((n_minimizers + 255) / 256); }
void launch_lmer_filter(const LmerFilterArgs& a, u32 fmt, const u8* bases, const uint2* planes, const u64* exc_pos, const u8* exc_val, u32 n_exc, hipStream_t s) {
    const u32 nb = lmer_filter_blocks(a.m1 - a.m0);
    if (!nb) return;
    if (fmt == FMT_ASCII) {
        const AsciiSrc src{bases};
        if (a.hpc) hipLaunchKernelGGL((lmer_filter_kernel<true, AsciiSrc>), dim3(nb), dim3(256), 0, s, a, src);
        else hipLaunchKernelGGL((lmer_filter_kernel<false, AsciiSrc>), dim3(nb), dim3(256), 0, s, a, src);
    } else {
        const PlaneSrc src{planes, exc_pos, exc_val, n_exc};
        if (a.hpc) hipLaunchKernelGGL((lmer_filter_kernel<true, PlaneSrc>), dim3(nb), dim3(256), 0, s, a, src);
        else hipLaunchKernelGGL((lmer_filter_kernel<false, PlaneSrc>), dim3(nb), dim3(256), 0, s, a, src);
    }
}

// hash_bound = floor(density * 2^64), saturating (src/read.rs:183)
u64 make_hash_bound(double density) {
    const double v = density * 18446744073709551616.0;
    return !(v > 0.0) ? 0 : (v >= 18446744073709551616.0 ? ~0ull : (u64)v);
}
