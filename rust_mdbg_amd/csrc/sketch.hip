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
//     5. output the tile's count enters a decoupled look-back scan over the tiles (one 64-bit state word per tile),
//               so the minimizers are written once, directly at their final, position-ordered place.
//   Tiles that hold a byte outside ACGT (N, lower case, garbage) or whose look-back window is one long
//   homopolymer take the generic exact walker inside the same kernel (slow_tile): exact, one thread per position.
#include "mdbg_dev.h"
#include "bs_core.h"

enum { FMT_ASCII = 0, FMT_PLANES = 1 };

struct SketchArgs {
    const u8* bases;             // FMT_ASCII: one byte per base
    const uint2* planes;         // FMT_PLANES: {plane0, plane1} per 32 bases, bit i = base i (include/mdbg_hip.h)
    u32 fmt;
    u64 n_bases;                 // positions >= n_bases do not exist
    const u64* offsets; u32 n_reads;
    const u32* bread;            // read containing the first staged base of tile t, [n_tiles + 2]
    u32 n_tiles;
    u64* tstate; u32* ticket;    // look-back state per tile (zeroed), tile ticket (zeroed)
    u64 out_base, out_cap; u64* out_hash; u32* out_pos; u32* out_read;
    u64* total_out;              // <- out_base + minimizers of the launch (written by the last tile)
    const u64* t4;               // 256 x {F4, R4} (bs_make_t4)
    const u8* tile_flags;        // FMT_PLANES: nonzero = an exception falls into the tile's staged range (null: none)
    const u64* exc_pos; const u8* exc_val; u32 n_exc;
    u32* err_flag; unsigned long long* slow_total;
    u32 read_base;               // slot index of the batch's first read in the resident store
    u64 bound; u32 l; u32 hpc; u32 btop;   // btop = top BS_B bits of the bound
    u32 force_slow;              // MDBG_FLAG_FORCE_GENERIC: every tile takes the generic exact walker
    u64* dbg;                    // diagnostic: per-tile phase timestamps [n_tiles][8] (null in production)
};

// largest r in [lo, hi] with off[r] <= p
__device__ inline u32 find_read(const u64* __restrict__ off, u32 lo, u32 hi, u64 p) {
    while (lo < hi) { u32 mid = lo + ((hi - lo + 1) >> 1); if (off[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}

// bread[t] = read containing the first staged base of tile t (t <= n_tiles + 1, clamped to the batch)
__global__ void bread_kernel(const u64* __restrict__ off, u32 n_reads, u64 n_bases, u32 n_entries, u32* __restrict__ bread) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_entries) return;
    int64_t p = (int64_t)t * TILE_STRIDE - HALO_BASES;
    if (p < 0) p = 0;
    if ((u64)p >= n_bases) p = n_bases ? (int64_t)n_bases - 1 : 0;
    bread[t] = find_read(off, 0, n_reads - 1, (u64)p);
}

// ---- input accessors of the generic walker -------------------------------------------------------------
struct AsciiSrc {
    const u8* b;
    __device__ u8 at(u64 q) const { return b[q]; }
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
constexpr int RW = TILE_RAW_WORDS;            // raw words staged per tile (halo included)
constexpr int HW = HALO_BASES / 32;           // leading halo words
constexpr int TT = TILE_THREADS;
constexpr int WPT = RW / TT;                  // raw words per thread
constexpr int DPAD = 4;                       // zero words in front of the dense stream (look-back of the first words)
constexpr int QCAP = 1024;                    // candidates per evaluation round
constexpr u64 TS_FLAG_A = 1ull << 62, TS_FLAG_P = 2ull << 62, TS_VAL = (1ull << 62) - 1;

struct __attribute__((aligned(16))) TileLds {
    u32 dense[2 * (DPAD + RW + 4)];           // dense word D: planes at [2 * (DPAD + D)], [.. + 1]
    u32 kw[RW];                               // keep mask of raw word w
    u16 rpre[RW + 8];                         // kept bases in front of raw word w; [RW] = all
    u16 dfirst[RW + 8];                       // raw word that holds dense position 32 * D
    union {
        u32 stage[2 * RW];                    // FMT_ASCII: half planes of the 16-base chunks (phase 1 only)
        struct { u32 cand[RW + 8]; u16 cpre[RW + 8]; } c;
    } a;
    union { u32 force[RW]; u16 list[QCAP]; } b;   // read starts (phases 1-2) | candidate list (phases 4-)
    u32 misc[32];                             // [0..4] scan scratch, [8] slow, [9] tile, [10] H, [11] Hh, [12..13] counts, [14,15] base, [16] next chunk
};
static_assert(sizeof(TileLds) * 5 <= 160 * 1024, "5 workgroups per CU");

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
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// decoupled look-back over the tiles of the launch: called by wave 0; returns the exclusive prefix (out_base included)
__device__ inline u64 lookback_publish(u64* tstate, u32 gt, u64 n, u64 out_base) {
    const int lane = threadIdx.x & 63;
    u64 excl = out_base;
    if (gt != 0) {
        if (lane == 0) __hip_atomic_store(&tstate[gt], TS_FLAG_A | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t j = (int64_t)gt;
        for (;;) {
            const int64_t idx = j - 1 - lane;
            u64 v = TS_FLAG_P;                                   // in front of tile 0: prefix 0
            if (idx >= 0) v = __hip_atomic_load(&tstate[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 flag = (u32)(v >> 62);
            const u64 notready = __ballot(flag == 0), isP = __ballot(flag == 2);
            const int firstP = isP ? __ffsll((unsigned long long)isP) - 1 : 64;
            const u64 need = firstP >= 63 ? ~0ull : ((2ull << firstP) - 1);
            if (notready & need) { __builtin_amdgcn_s_sleep(2); continue; }
            u64 contrib = lane <= firstP ? (v & TS_VAL) : 0;
            for (int d = 32; d; d >>= 1) contrib += __shfl_xor(contrib, d, 64);
            excl += contrib;
            if (firstP < 64) break;
            j -= 64;
        }
    }
    if (lane == 0) __hip_atomic_store(&tstate[gt], TS_FLAG_P | (excl - out_base + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// ---- generic exact path for one tile (inside the tile kernel) --------------------------------------------
template <bool HPC, class Src>
__device__ void slow_tile(const SketchArgs& a, const Src& src, u32 gt, TileLds& S) {
    const int tid = threadIdx.x;
    const u64 t_lo = (u64)gt * TILE_STRIDE;
    u64 t_hi = t_lo + TILE_STRIDE; if (t_hi > a.n_bases) t_hi = a.n_bases;
    const u32 rl = a.bread[gt], rh_ = a.bread[gt + 2];
    const u64 first_base = a.offsets[0];
    u32* tmp = S.misc;
    if (tid == 0) atomicAdd(a.slow_total, 1ull);
    u64 base = 0;
    for (int pass = 0; pass < 2; ++pass) {
        u32 running = 0;
        for (u64 p0 = t_lo; p0 < t_hi; p0 += TT) {
            const u64 p = p0 + tid;
            u32 sel = 0; u64 hash = 0, start = 0; u32 r = 0; u64 rlo = 0;
            if (p < t_hi && p >= first_base) {
                r = find_read(a.offsets, rl, rh_, p);
                rlo = a.offsets[r];
                if (pass == 0) { const u8 c = src.at(p); if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1; }   // the host applies the length rule
                if (kept_at<HPC>(src, rlo, p) && walk_lmer<HPC>(src, rlo, p, a.l, start, hash) && hash <= a.bound) sel = 1;
            }
            if (pass == 0) { running += sel; continue; }
            u32 total;
            const u32 rank = block_excl_scan_256(sel, tmp, total);
            if (sel) {
                const u64 idx = base + running + rank;
                if (idx < a.out_cap) { a.out_hash[idx] = hash; a.out_pos[idx] = (u32)(start - rlo); a.out_read[idx] = r + a.read_base; }
            }
            running += total;
        }
        if (pass == 0) {
            u32 total;
            (void)block_excl_scan_256(running, tmp, total);
            if (tid < 64) {
                const u64 excl = lookback_publish(a.tstate, gt, total, a.out_base);
                if (tid == 0) { S.misc[14] = (u32)excl; S.misc[15] = (u32)(excl >> 32); if (gt == a.n_tiles - 1) *a.total_out = excl + total; }
            }
            __syncthreads();
            base = (u64)S.misc[14] | ((u64)S.misc[15] << 32);
        }
    }
}

// ---- fast tile kernel ---------------------------------------------------------------------------------------------
struct CandOut { u64 hash; u32 pos, read; };

template <int L>
__global__ __launch_bounds__(TT) void sketch_bs_kernel(SketchArgs a) {
    __shared__ TileLds S;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { S.misc[9] = atomicAdd(a.ticket, 1u); S.misc[8] = a.force_slow; S.misc[11] = 0; }
    for (int i = tid; i < 2 * (DPAD + RW + 4); i += TT) S.dense[i] = 0;
    for (int i = tid; i < RW; i += TT) S.b.force[i] = 0;
    __syncthreads();
    const u32 gt = S.misc[9];
#define MDBG_STAMP(i) do { if (a.dbg && tid == 0) a.dbg[(size_t)gt * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    MDBG_STAMP(0);
    const int64_t raw0 = (int64_t)gt * TILE_STRIDE - HALO_BASES;      // first staged raw position (negative for tile 0)
    const int64_t nb = (int64_t)a.n_bases;
    const bool hpc = a.hpc != 0;
    const u32 rl = a.bread[gt], rh_ = a.bread[gt + 2];
    const int64_t first_base = (int64_t)a.offsets[0];      // positions in front of it belong to no read

    // ---- phase 1: read starts, load, planes ---------------------------------------------------------------------
    for (u32 r = rl + tid; r <= rh_ && r < a.n_reads; r += TT) {
        const int64_t rel = (int64_t)a.offsets[r] - raw0;
        if (rel >= 0 && rel < RW * 32) atomicOr(&S.b.force[rel >> 5], 0x80000000u >> (rel & 31));
    }
    u32 x0[WPT], x1[WPT], pv0 = 0, pv1 = 0;          // my raw words (MSB first); pv*: bit 0 = the base in front of them
    const bool interior = raw0 >= 0 && raw0 + RW * 32 <= nb;
    if (a.fmt == FMT_ASCII) {
        u32 bad_any = 0;
        constexpr int CPT = RW * 2 / TT;              // 16-base chunks per thread
        if (interior) {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = (const u32x4*)(a.bases + raw0);
            uint4 v[CPT];
#pragma unroll
            for (int g = 0; g < CPT; ++g) { const u32x4 q = __builtin_nontemporal_load(src + tid + TT * g); v[g] = make_uint4(q.x, q.y, q.z, q.w); }
#pragma unroll
            for (int g = 0; g < CPT; ++g) S.a.stage[tid + TT * g] = ascii16_to_hp(v[g], bad_any);
        } else {
#pragma unroll 1
            for (int g = 0; g < CPT; ++g) {
                const int ci = tid + TT * g;
                const int64_t pos = raw0 + 16 * (int64_t)ci;
                u32 w4[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};     // positions outside the batch read as 'A' (masked below)
                if (pos >= 0 && pos + 16 <= nb) { const uint4 q = *(const uint4*)(a.bases + pos); w4[0] = q.x; w4[1] = q.y; w4[2] = q.z; w4[3] = q.w; }
                else for (int i = 0; i < 16; ++i) if (pos + i >= 0 && pos + i < nb) w4[i >> 2] = (w4[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((u32)a.bases[pos + i] << (8 * (i & 3)));
                S.a.stage[ci] = ascii16_to_hp(make_uint4(w4[0], w4[1], w4[2], w4[3]), bad_any);
            }
        }
        if (bad_any) {                               // some byte of my chunks is not one of ACGT: the tile takes the generic path
            S.misc[8] = 1;
            for (int g = 0; g < CPT; ++g) {
                const int64_t pos = raw0 + 16 * (int64_t)(tid + TT * g);
                for (int i = 0; i < 16; ++i) {
                    const int64_t q = pos + i;
                    if (q >= 0 && q < nb) { const u8 c = a.bases[q]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1; }
                }
            }
        }
        __syncthreads();
        const uint4* st4 = (const uint4*)(S.a.stage + 2 * WPT * tid);
#pragma unroll
        for (int i = 0; i < WPT / 2; ++i) {
            const uint4 h = st4[i];
            x0[2 * i] = (h.x & 0xFFFF0000u) | (h.y >> 16);     x1[2 * i] = (h.x << 16) | (h.y & 0xFFFFu);
            x0[2 * i + 1] = (h.z & 0xFFFF0000u) | (h.w >> 16); x1[2 * i + 1] = (h.z << 16) | (h.w & 0xFFFFu);
        }
        if (tid) { const u32 hp = S.a.stage[2 * WPT * tid - 1]; pv0 = hp >> 16; pv1 = hp; }
    } else {
        const int64_t pi0 = raw0 / 32 + (int64_t)WPT * tid;       // raw0 is a multiple of 32 (also when negative)
        const int64_t n_pairs = (nb + 31) >> 5;
        uint2 pr[WPT];
        if (interior) {
            const uint4* src = (const uint4*)(a.planes + pi0);
#pragma unroll
            for (int i = 0; i < WPT / 2; ++i) { const uint4 q = src[i]; pr[2 * i] = make_uint2(q.x, q.y); pr[2 * i + 1] = make_uint2(q.z, q.w); }
        } else {
#pragma unroll
            for (int i = 0; i < WPT; ++i) { const int64_t pi = pi0 + i; pr[i] = (pi >= 0 && pi < n_pairs) ? a.planes[pi] : make_uint2(0u, 0u); }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) { x0[i] = __brev(pr[i].x); x1[i] = __brev(pr[i].y); }
        if (tid && pi0 - 1 >= 0 && pi0 - 1 < n_pairs) { const uint2 q = a.planes[pi0 - 1]; pv0 = q.x >> 31; pv1 = q.y >> 31; }
        if (tid == 0 && a.tile_flags && a.tile_flags[gt]) S.misc[8] = 1;
    }
    MDBG_STAMP(1);

    // ---- phase 2: keep masks, compaction, dense stream ----------------------------------------------------------------
    u32 kw[WPT], n_kept[WPT], mine = 0;
    {
        const int64_t lo = first_base - raw0, hi = nb - raw0;        // existing positions, tile-relative
#pragma unroll
        for (int i = 0; i < WPT; ++i) {                      // keep masks first: they look at the neighbouring raw word
            const int w = WPT * tid + i;
            const u32 vm = interior && lo <= 0 ? 0xFFFFFFFFu : range_mask(lo - 32 * (int64_t)w, hi - 32 * (int64_t)w);
            u32 k = vm;
            if (hpc) {
                const u32 d0 = bs_alignbit(i ? x0[i - 1] : pv0, x0[i], 1), d1 = bs_alignbit(i ? x1[i - 1] : pv1, x1[i], 1);
                k = ((x0[i] ^ d0) | (x1[i] ^ d1) | S.b.force[w]) & vm;
                if (w == 0) k |= 0x80000000u & vm;           // nothing staged in front of the first position
            }
            kw[i] = k; n_kept[i] = bs_popc(k); mine += n_kept[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            if (hpc || kw[i] != 0xFFFFFFFFu) bs_compress2(kw[i], x0[i], x1[i]);
        }
    }
    u32 H;
    u32 off = block_excl_scan_256(mine, S.misc, H);
    if (tid == HW / WPT) S.misc[11] = off;                   // kept bases of the halo words
    if (tid == TT - 1) S.rpre[RW] = (u16)H;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int w = WPT * tid + i;
        S.kw[w] = kw[i]; S.rpre[w] = (u16)off;
        const u32 n = n_kept[i];
        if (n) {
            const u32 wi = off >> 5, s = off & 31;
            const u32 d_first = (off + 31) >> 5;
            if (32 * d_first < off + n) S.dfirst[d_first] = (u16)w;
            unsigned long long* dst = (unsigned long long*)(S.dense + 2 * (DPAD + wi));
            atomicOr(dst, (unsigned long long)(x0[i] >> s) | ((unsigned long long)(x1[i] >> s) << 32));
            if (s + n > 32) atomicOr(dst + 1, (unsigned long long)bs_alignbit(x0[i], 0u, s) | ((unsigned long long)bs_alignbit(x1[i], 0u, s) << 32));
        }
        off += n;
    }
    __syncthreads();
    MDBG_STAMP(2);
    const u32 Hh = S.misc[11];
    const bool true_start = raw0 <= first_base;       // the stream begins inside this tile: nothing to look back at
    if (S.misc[8] || (!true_start && Hh < (u32)L)) {
        // N / foreign bytes, or the look-back window is one long homopolymer: exact generic walker for the whole tile
        __syncthreads();
        if (a.fmt == FMT_ASCII) { AsciiSrc src{a.bases}; if (hpc) slow_tile<true>(a, src, gt, S); else slow_tile<false>(a, src, gt, S); }
        else { PlaneSrc src{a.planes, a.exc_pos, a.exc_val, a.tile_flags && a.tile_flags[gt] ? a.n_exc : 0u}; if (hpc) slow_tile<true>(a, src, gt, S); else slow_tile<false>(a, src, gt, S); }
        return;
    }

    // ---- phase 3: bit-sliced filter over the dense stream ---------------------------------------------------------
    {
        u32 bmask[BS_B];
#pragma unroll
        for (int i = 0; i < BS_B; ++i) bmask[i] = ((a.btop >> (BS_B - 1 - i)) & 1u) ? 0xFFFFFFFFu : 0u;
        const u32 n_out = H ? ((H + BS_B - 2) >> 5) + 1 : 0;          // words of the candidate plane (shifted by BS_B - 1)
        const u32 n_steps = (n_out + 62) / 63;
        for (u32 st = wv; st < n_steps; st += TT / 64) {
            const int D = (int)(63 * st) + lane - 1;                  // lane 0 recomputes the word before the step's first
            const u32* dw = S.dense + 2 * (DPAD + (D < RW + 3 ? D : RW + 3));     // words past the stream are zero; their results are dropped
            const uint2 c = *(const uint2*)dw, p = *(const uint2*)(dw - 2);
            uint2 q = make_uint2(0u, 0u);
            if (L + BS_B - 2 >= 32) q = *(const uint2*)(dw - 4);
            u32 W[BS_B], Wp[BS_B], inv;
            bs_strand_planes<L, true>(c.x, c.y, p.x, p.y, q.x, q.y, W, inv);
#pragma unroll
            for (int i = 0; i < BS_B; ++i) Wp[i] = dpp_wave_shr1(W[i]);
            u32 cand = bs_strand_compare<true>(W, Wp, inv, bmask);
            bs_strand_planes<L, false>(c.x, c.y, p.x, p.y, q.x, q.y, W, inv);
#pragma unroll
            for (int i = 0; i < BS_B; ++i) Wp[i] = dpp_wave_shr1(W[i]);
            cand |= bs_strand_compare<false>(W, Wp, inv, bmask);
            if (lane && (u32)D < n_out) S.a.c.cand[D] = cand;
        }
    }
    __syncthreads();
    MDBG_STAMP(3);

    // ---- phase 4: owned candidates, exact evaluation, output -----------------------------------------------------------
    // candidate plane coordinate x = e + BS_B - 1; owned END positions e in [max(Hh, L-1), H)
    const u32 e_lo = Hh > (u32)(L - 1) ? Hh : (u32)(L - 1);
    const int nw = tid == TT - 1 ? WPT + 1 : WPT;                     // words 4*tid .. ; the last thread also takes word RW
    auto count_words = [&](bool mask_range) -> u32 {                  // cpre[] <- exclusive candidate counts per word; returns the total
        u32 cnt = 0, cw[WPT + 1];
        for (int i = 0; i < nw; ++i) {
            const int D = WPT * tid + i;
            u32 w = S.a.c.cand[D];
            if (mask_range) {
                const u32 n_out = H ? ((H + BS_B - 2) >> 5) + 1 : 0;
                w = (u32)D < n_out ? w & range_mask((int64_t)e_lo + BS_B - 1 - 32 * (int64_t)D, (int64_t)H + BS_B - 1 - 32 * (int64_t)D) : 0u;
                S.a.c.cand[D] = w;
            }
            cw[i] = bs_popc(w); cnt += cw[i];
        }
        u32 total;
        u32 o = block_excl_scan_256(cnt, S.misc, total);
        for (int i = 0; i < nw; ++i) { S.a.c.cpre[WPT * tid + i] = (u16)o; o += cw[i]; }
        return total;
    };
    // one round of candidates = whole words, at most QCAP candidates, starting at rank c0; list[] <- their END positions.
    // returns the rank after the round
    auto build_list = [&](u32 c0) -> u32 {
        if (tid == 0) S.misc[16] = c0;
        __syncthreads();
        u32 hi = c0;
        for (int i = 0; i < nw; ++i) {
            const int D = WPT * tid + i;
            u32 w = S.a.c.cand[D];
            u32 r = S.a.c.cpre[D];
            const u32 n = bs_popc(w);
            if (n == 0 || r < c0 || r + n > c0 + QCAP) continue;
            hi = r + n;
            while (w) { const u32 b = (u32)__clz(w); w &= ~(0x80000000u >> b); S.b.list[r - c0] = (u16)(32 * D + b - (BS_B - 1)); ++r; }
        }
        if (hi > c0) atomicMax(&S.misc[16], hi);
        __syncthreads();
        return S.misc[16];
    };
    auto dense_to_raw = [&](u32 r) -> u32 {                            // tile-relative raw position of dense position r
        u32 w = S.dfirst[r >> 5];
        while (S.rpre[w + 1] <= r) ++w;
        return 32 * w + bs_select_msb(S.kw[w], r - S.rpre[w]);
    };
    auto eval = [&](u32 e, CandOut& o) -> bool {                       // exact: src/read.rs:196-208 for the l-mer ending at dense position e
        const u32 wi = e >> 5, s = e & 31;
        const u32* dw = S.dense + 2 * (DPAD + wi);
        const u32 v0 = bs_alignbit(dw[-2], dw[0], 31 - s), v1 = bs_alignbit(dw[-1], dw[1], 31 - s);
        const u64 h = bs_exact_hash(v0, v1, L, a.t4);
        if (h > a.bound) return false;
        const int64_t abs_end = raw0 + dense_to_raw(e);
        const u32 r = find_read(a.offsets, rl, rh_, (u64)abs_end);
        const int64_t q0 = (int64_t)a.offsets[r];
        const u32 sd = e - (u32)(L - 1);
        if (q0 > raw0) {                                               // the read starts inside the staged range: the l-mer must not cross it
            const u32 rel = (u32)(q0 - raw0), w = rel >> 5, b = rel & 31;
            const u32 ds = S.rpre[w] + (b ? bs_popc(S.kw[w] >> (32 - b)) : 0u);
            if (sd < ds) return false;
        }
        o.hash = h; o.pos = (u32)(raw0 + dense_to_raw(sd) - q0); o.read = r + a.read_base;
        return true;
    };

    const u32 n_cand = count_words(true);
    CandOut keep[QCAP / TT]; u32 keep_ok = 0;
    const bool one_round = n_cand <= QCAP;
    // pass A: validate every candidate, clear the bits of the ones that fail (rounds are whole words: later rounds unaffected)
    for (u32 c0 = 0; c0 < n_cand;) {
        const u32 c1 = build_list(c0);
#pragma unroll
        for (int i = 0; i < QCAP / TT; ++i) {
            const u32 j = tid + TT * i;
            if (j < c1 - c0) {
                const u32 e = S.b.list[j];
                CandOut o;
                const bool ok = eval(e, o);
                if (ok && one_round) { keep[i] = o; keep_ok |= 1u << i; }
                if (!ok) { const u32 x = e + BS_B - 1; atomicAnd(&S.a.c.cand[x >> 5], ~(0x80000000u >> (x & 31))); }
            }
        }
        __syncthreads();
        c0 = c1;
    }
    MDBG_STAMP(4);
    const u32 nv = n_cand ? count_words(false) : 0;
    if (tid < 64) {
        const u64 excl = lookback_publish(a.tstate, gt, nv, a.out_base);
        if (tid == 0) { S.misc[14] = (u32)excl; S.misc[15] = (u32)(excl >> 32); if (gt == a.n_tiles - 1) *a.total_out = excl + nv; }
    }
    __syncthreads();
    const u64 base = (u64)S.misc[14] | ((u64)S.misc[15] << 32);
    if (one_round) {
        // ranks among the survivors: position of the candidate's bit in the refined bitmap
#pragma unroll
        for (int i = 0; i < QCAP / TT; ++i) if ((keep_ok >> i) & 1u) {
            const u32 e = S.b.list[tid + TT * i], x = e + BS_B - 1, D = x >> 5, b = x & 31;
            const u64 idx = base + S.a.c.cpre[D] + (b ? bs_popc(S.a.c.cand[D] >> (32 - b)) : 0u);
            if (idx < a.out_cap) { a.out_hash[idx] = keep[i].hash; a.out_pos[idx] = keep[i].pos; a.out_read[idx] = keep[i].read; }
        }
    } else {
        for (u32 c0 = 0; c0 < nv;) {
            const u32 c1 = build_list(c0);
#pragma unroll
            for (int i = 0; i < QCAP / TT; ++i) {
                const u32 j = tid + TT * i;
                if (j < c1 - c0) {
                    CandOut o;
                    if (eval(S.b.list[j], o)) { const u64 idx = base + c0 + j; if (idx < a.out_cap) { a.out_hash[idx] = o.hash; a.out_pos[idx] = o.pos; a.out_read[idx] = o.read; } }
                }
            }
            __syncthreads();
            c0 = c1;
        }
    }
    MDBG_STAMP(5);
#undef MDBG_STAMP
}

// per-read offsets into the ordered minimizer arrays: off[slot] = first i with mread[i] >= slot, for the batch's
// slots [slot0, slot0 + n_reads]; mread holds absolute slot indices and is sorted.
__global__ void read_offsets_kernel(const u32* __restrict__ mread, u64 m0, u64 m1, u32 slot0, u32 n_reads, u64* __restrict__ off) {
    const u64 i = m0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > m1) return;
    const int64_t prev = (i == m0) ? (int64_t)slot0 - 1 : (int64_t)mread[i - 1];
    const int64_t cur = (i == m1) ? (int64_t)slot0 + n_reads : (int64_t)mread[i];
    for (int64_t r = prev + 1; r <= cur; ++r) off[r] = i;
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
__global__ void tile_flags_kernel(const u64* __restrict__ exc_pos, u32 n_exc, u32 n_tiles, u8* __restrict__ flags) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_exc) return;
    const u64 p = exc_pos[i];
    const u64 t = p / TILE_STRIDE;
    if (t < n_tiles) flags[t] = 1;
    if (t + 1 < n_tiles && p + HALO_BASES >= (t + 1) * (u64)TILE_STRIDE) flags[t + 1] = 1;      // falls into the next tile's look-back window
}

// ---- host launchers -------------------------------------------------------------------------------
void launch_bread(const u64* offsets, u32 n_reads, u64 n_bases, u32 n_tiles, u32* bread, hipStream_t s) {
    const u32 n = n_tiles + 2;
    hipLaunchKernelGGL(bread_kernel, dim3((n + 255) / 256), dim3(256), 0, s, offsets, n_reads, n_bases, n, bread);
}
void launch_tile_flags(const u64* exc_pos, u32 n_exc, u32 n_tiles, u8* flags, hipStream_t s) {
    if (n_exc) hipLaunchKernelGGL(tile_flags_kernel, dim3((n_exc + 255) / 256), dim3(256), 0, s, exc_pos, n_exc, n_tiles, flags);
}
void launch_alphabet_rule(const SketchArgs& a, unsigned long long* which, hipStream_t s) {
    if (!a.n_reads) return;
    const dim3 g((a.n_reads + 3) / 4), b(256);
    if (a.fmt == FMT_ASCII) hipLaunchKernelGGL(alphabet_rule_kernel<AsciiSrc>, g, b, 0, s, AsciiSrc{a.bases}, a.offsets, a.n_reads, a.l, a.hpc, which);
    else hipLaunchKernelGGL(alphabet_rule_kernel<PlaneSrc>, g, b, 0, s, PlaneSrc{a.planes, a.exc_pos, a.exc_val, a.n_exc}, a.offsets, a.n_reads, a.l, a.hpc, which);
}

template <int L> static void launch_bs(const SketchArgs& a, hipStream_t s) { hipLaunchKernelGGL(sketch_bs_kernel<L>, dim3(a.n_tiles), dim3(TT), 0, s, a); }
// one launch covers the whole batch (launch boundaries would only re-synchronise the workgroups' phases)
void launch_sketch(const SketchArgs& a, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (!a.n_tiles) return;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    switch (a.l) {
#define MDBG_L(n) case n: launch_bs<n>(a, s); break;
        MDBG_L(2) MDBG_L(3) MDBG_L(4) MDBG_L(5) MDBG_L(6) MDBG_L(7) MDBG_L(8) MDBG_L(9) MDBG_L(10) MDBG_L(11) MDBG_L(12) MDBG_L(13)
        MDBG_L(14) MDBG_L(15) MDBG_L(16) MDBG_L(17) MDBG_L(18) MDBG_L(19) MDBG_L(20) MDBG_L(21) MDBG_L(22) MDBG_L(23) MDBG_L(24)
        MDBG_L(25) MDBG_L(26) MDBG_L(27) MDBG_L(28) MDBG_L(29) MDBG_L(30) MDBG_L(31) MDBG_L(32)
#undef MDBG_L
        default: break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
}

void launch_read_offsets(const u32* mread, u64 m0, u64 m1, u32 slot0, u32 n_reads, u64* off, hipStream_t s) {
    const u64 n = m1 - m0 + 1;
    hipLaunchKernelGGL(read_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mread, m0, m1, slot0, n_reads, off);
}

// hash_bound = floor(density * 2^64), saturating (src/read.rs:183)
u64 make_hash_bound(double density) {
    const double v = density * 18446744073709551616.0;
    return !(v > 0.0) ? 0 : (v >= 18446744073709551616.0 ? ~0ull : (u64)v);
}
