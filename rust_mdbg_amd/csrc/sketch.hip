// sketch.hip — HPC + ntHash + density filter, fused (gfx950).
//
// Replaces Read::encode_rle + Read::extract_density (rust-mdbg src/read.rs:157-211) and the nthash
// crate's NtHashIterator for a whole batch of reads.  Layout and algorithm: DESIGN.md §"sketch kernel".
//
//   sketch_tile_kernel   fast path: one workgroup per 64 Ki raw bases of the concatenated batch.
//                        ASCII is read once with coalesced 16-byte loads, packed to 2-bit codes in LDS,
//                        each lane then rolls 32-bit partial fwd/rev ntHash values over its own 256-base
//                        segment (HPC = "push only when the code changes"), flags candidates in a bitmap,
//                        and an exact 64-bit fix-up (walk back over l run starts) validates every candidate.
//   slow_tile_kernel     exact generic path straight from ASCII (N, invalid bytes, l > 14, dense tiles).
//   gather_kernel        squeezes the per-tile candidate slabs into the final, position-ordered arrays.
#include "mdbg_dev.h"

struct TileArgs {
    const u8* bases; u64 n_bases; const u64* offsets; u32 n_reads;
    const u32* bread;            // read containing the first base of tile t, [n_tiles_total + 1]
    u64 tile0; u32 n_tiles;      // this launch covers tiles [tile0, tile0 + n_tiles)
    Rec* slab;                   // [n_tiles][QCAP]
    u32* n_cand; u32* n_valid;   // per tile of the launch
    u32* slow_list; u32* slow_count;
    u32* err_flag;               // set when a byte outside ACGTN is seen
    const u32* tbl;              // device copy of SketchConsts::tbl
    u32 read_base;               // slot index of the batch's first read in the resident store
    u64* dbg;                    // diagnostic: per-tile phase timestamps [n_tiles][8] (null in production)
    SketchConsts c;
};

// largest r in [lo, hi] with off[r] <= p
__device__ inline u32 find_read(const u64* __restrict__ off, u32 lo, u32 hi, u64 p) {
    while (lo < hi) { u32 mid = lo + ((hi - lo + 1) >> 1); if (off[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ void bread_kernel(const u64* __restrict__ off, u32 n_reads, u64 n_tiles_total, u32* __restrict__ bread) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles_total) return;
    bread[t] = (t == n_tiles_total) ? n_reads - 1 : find_read(off, 0, n_reads - 1, t * (u64)TILE);
}

// ---- exact generic walker on ASCII (src/read.rs:157-174 semantics) -----------------------------
template <bool HPC>
__device__ inline bool kept_ascii(const u8* __restrict__ b, u64 rlo, u64 p) {
    if (!HPC || p == rlo) return true;
    u8 c = b[p];
    return !(c == b[p - 1] && in_hpc_set(c));
}
// l-mer whose LAST HPC base is the run starting at p.  false: fewer than l HPC bases precede p in the read.
template <bool HPC>
__device__ inline bool walk_lmer_ascii(const u8* __restrict__ b, u64 rlo, u64 p, u32 l, u64& start, u64& hash) {
    u64 q = p, fh = 0, rh = 0;
    for (int j = (int)l - 1;; --j) {
        u8 c = b[q];
        fh ^= rol64(nt_h_ascii(c), l - 1 - j);
        rh ^= rol64(nt_rc_ascii(c), j);
        if (j == 0) break;
        if (q == rlo) return false;
        u64 q2 = q - 1;
        if (HPC) { u8 c2 = b[q2]; if (in_hpc_set(c2)) while (q2 > rlo && b[q2 - 1] == c2) --q2; }
        q = q2;
    }
    start = q; hash = fh < rh ? fh : rh;
    return true;
}

// ---- generic exact tile kernel -------------------------------------------------------------------
// WRITE=false: counts the selected minimizers of every slow tile (n_valid).  WRITE=true: writes them at tile_base.
template <bool HPC, bool WRITE>
__global__ __launch_bounds__(256) void slow_tile_kernel(TileArgs a, const u64* __restrict__ tile_base,
                                                        u64* __restrict__ out_hash, u32* __restrict__ out_pos,
                                                        u32* __restrict__ out_read, u64 out_cap) {
    __shared__ u32 tmp[8];
    const u32 n_slow = *a.slow_count;
    for (u32 s = blockIdx.x; s < n_slow; s += gridDim.x) {
    const u32 t = a.slow_list[s];
    const u64 gt = a.tile0 + t;
    const u64 tile_start = gt * (u64)TILE;
    const u64 tile_end = tile_start + TILE < a.n_bases ? tile_start + TILE : a.n_bases;
    const u32 rl = a.bread[gt], rh_ = a.bread[gt + 1];
    u32 running = 0;
    for (u64 base = tile_start; base < tile_end; base += 256) {
        const u64 p = base + threadIdx.x;
        u32 sel = 0; u64 hash = 0, start = 0; u32 r = 0; u64 rlo = 0;
        if (p < tile_end) {
            r = find_read(a.offsets, rl, rh_, p);
            rlo = a.offsets[r];
            if (!WRITE) { const u8 c = a.bases[p]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1; }   // host decides (read length rule)
            if (p >= rlo && kept_ascii<HPC>(a.bases, rlo, p) && walk_lmer_ascii<HPC>(a.bases, rlo, p, a.c.l, start, hash) && hash <= a.c.bound) sel = 1;
        }
        u32 total;
        u32 rank = block_excl_scan_256(sel, tmp, total);
        if (WRITE && sel) {
            u64 idx = tile_base[t] + running + rank;
            if (idx < out_cap) { out_hash[idx] = hash; out_pos[idx] = (u32)(start - rlo); out_read[idx] = r + a.read_base; }
        }
        running += total;
    }
    if (!WRITE && threadIdx.x == 0) a.n_valid[t] = running;
    __syncthreads();
    }
}

// marks every tile of the launch slow (l > FAST_MAX_L or forced)
__global__ void all_slow_kernel(u32 n_tiles, u32* __restrict__ n_cand, u32* __restrict__ slow_list, u32* __restrict__ slow_count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles) { n_cand[t] = SLOW_MARK; slow_list[t] = t; }
    if (t == 0) *slow_count = n_tiles;
}

// ---- fast tile kernel ------------------------------------------------------------------------------
// LDS (dwords): 2-bit codes of the tile + halo | candidate list (u16) | rotate tables of the fix-up | transition table
constexpr int LDS_CODES = (TILE_THREADS + 1) * SEG_WORDS;   // logical word j (>= -HALO/16) lives at index j + SEG_WORDS
constexpr int LDS_LIST = QCAP / 2;
constexpr int LDS_RT = MDBG_MAX_L_DEV * 4 * 4;              // {rol(h[c], l-1-j), rol(rc[c], j)} for j < 32, c < 4: 16 B each
constexpr int LDS_TBL = 36;                                 // 16 x {XF, XR} + one all-zero entry
constexpr int LDS_MISC = 16;
constexpr int LDS_TOTAL = LDS_CODES + LDS_LIST + LDS_RT + LDS_TBL + LDS_MISC;   // 5700 dwords = 22.8 KB -> 7 workgroups per CU
static_assert(LDS_TOTAL * 4 * 7 <= 160 * 1024, "7 workgroups per CU");

__device__ inline u32 pack16(uint4 v, u32& bad) {
    u32 out = 0;
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 s = (w[i] >> 1) & 0x03030303u;
        u32 recon = __builtin_amdgcn_perm(0u, 0x47544341u /* 'A','C','T','G' by code */, s);
        bad |= recon ^ w[i];
        u32 byte = __builtin_amdgcn_udot4(s, 0x40100401u, 0u, false);   // c0 + 4 c1 + 16 c2 + 64 c3
        out |= byte << (8 * i);
    }
    return out;
}

// acc = 2*acc + (lane's bit of mask): one VALU instruction (v_addc with the carry-in taken from an SGPR pair)
__device__ inline u32 shift_in_bit(u32 acc, u64 mask) {
    u32 out; u64 carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(carry_out) : "v"(acc), "s"(mask));
    return out;
}

// Integer VALU issue rates on gfx950 (profiles/r01_g_valu_rates.txt): v_and/or/xor, v_add_u32, v_lshrrev and v_bitop3 with
// VGPR / inline-constant operands issue in 2 cycles per wave64 instruction; v_lshlrev, v_bfe, v_lshl_or and any instruction
// with an SGPR source take 4.  The few wrappers below pin the 2-cycle form where the compiler would pick a 4-cycle one
// (it counts instructions, not issue cycles); everything else in the step is left to the compiler.
__device__ __forceinline__ u32 v_and24(u32 x) { u32 d; asm("v_and_b32 %0, 24, %1" : "=v"(d) : "v"(x)); return d; }
__device__ __forceinline__ u32 v_shr2(u32 x) { u32 d; asm("v_lshrrev_b32 %0, 2, %1" : "=v"(d) : "v"(x)); return d; }
__device__ __forceinline__ u32 v_shr(u32 x, u32 sh) { u32 d; asm("v_lshrrev_b32 %0, %1, %2" : "=v"(d) : "v"(sh), "v"(x)); return d; }
__device__ __forceinline__ u32 v_dbl(u32 x) { u32 d; asm("v_add_u32 %0, %1, %1" : "=v"(d) : "v"(x)); return d; }                  // x << 1
__device__ __forceinline__ u32 v_and_or(u32 x, u32 y, u32 z) { u32 d; asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; }   // (x & y) | z
__device__ __forceinline__ u32 v_and(u32 x, u32 y) { u32 d; asm("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }

struct RollState { u32 hist, G, R, prev8; };       // prev8: code of the previous raw base << 3 (32: none)
struct RollConsts { u32 hs_shift, k60, maskR; };   // kept in VGPRs: an SGPR source would halve the issue rate of its instruction

// Eight steps of the rolling partial hashes over 8 two-bit codes (ws: code i at bits [4+2i : 3+2i]), branch-free: the code
// history / table addresses of all 8 steps first (a short ALU chain), then the 8 LDS reads back to back, then the G/R
// chains.  Lanes whose base is not a run start compute the same values and discard them with v_cndmask: with 64 lanes
// some lane always pushes, so an exec-mask branch would never be skipped and would serialise every ds_read's latency.
// Returns 8 bits, step 0 in bit 7: candidate flags (EMIT) or kept flags (!EMIT).
template <bool HPC, bool EMIT>
__device__ inline u32 roll8(RollState& st, u32 ws, const u32* tbl, const RollConsts& K, u32 thrF, u32 thrR) {
    u32 ad[8]; u64 kpm[8]; bool kp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 c8_ = v_and24(ws);
        ws = v_shr2(ws);
        kp[i] = !HPC || c8_ != st.prev8;
        kpm[i] = __builtin_amdgcn_ballot_w64(kp[i]);
        ad[i] = v_and_or(v_shr(st.hist, K.hs_shift), K.k60, c8_);          // (out code << 5) | (in code << 3)
        const u32 hn = (st.hist << 2) | c8_;
        st.hist = __builtin_unpredictable(kp[i]) ? hn : st.hist;
        st.prev8 = c8_;
    }
    uint2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = *(const uint2*)((const char*)tbl + ad[i]);
    u32 bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 Gn = v_dbl(st.G) ^ x[i].x, Rn = (st.R >> 1) ^ x[i].y;
        st.G = __builtin_unpredictable(kp[i]) ? Gn : st.G;
        st.R = __builtin_unpredictable(kp[i]) ? Rn : st.R;
        if (EMIT) {
            // a run continuation keeps G/R, so it repeats its predecessor's verdict: harmless, the fix-up rejects
            // positions that are not run starts
            const u64 cm = __builtin_amdgcn_ballot_w64(st.G <= thrF) | __builtin_amdgcn_ballot_w64(v_and(st.R, K.maskR) <= thrR);
            bits = shift_in_bit(bits, cm);
        } else {
            bits = shift_in_bit(bits, kpm[i]);
        }
    }
    return bits;
}

template <bool HPC>
__global__ __launch_bounds__(TILE_THREADS, 7) void sketch_tile_kernel(TileArgs a) {
    __shared__ __attribute__((aligned(16))) u32 lds[LDS_TOTAL];
    u32* const codes = lds + SEG_WORDS;                  // codes[j], j >= -HALO/16
    u16* const list = (u16*)(lds + LDS_CODES);
    u64* const rt = (u64*)(lds + LDS_CODES + LDS_LIST);  // rt[(j*4 + c)*2 + {0,1}]
    u32* const tbl = lds + LDS_CODES + LDS_LIST + LDS_RT;
    u32* const misc = tbl + LDS_TBL;                     // [0..4] scan tmp, [8] slow flag

    const int tid = threadIdx.x;
    const u32 t = blockIdx.x;
    const u64 gt = a.tile0 + t;
    const int64_t tile_start = (int64_t)(gt * (u64)TILE);
    const int64_t nb = (int64_t)a.n_bases;
    const u32 l = a.c.l;
#define MDBG_STAMP(i) do { if (a.dbg && tid == 0) a.dbg[(size_t)t * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    MDBG_STAMP(0);

    if (tid < 32) tbl[tid] = a.tbl[tid];
    if (tid >= 32 && tid < 36) tbl[tid] = 0;
    if (tid == 36) misc[8] = 0;
    if (tid >= 64 && tid < 64 + MDBG_MAX_L_DEV * 4) {
        const u32 e = tid - 64, j = e >> 2, c = e & 3;
        const u64 h4[4] = {NT_SEED_A, NT_SEED_C, NT_SEED_T, NT_SEED_G}, r4[4] = {NT_SEED_T, NT_SEED_G, NT_SEED_A, NT_SEED_C};
        u64 hv = h4[0], rv = r4[0];
        if (c == 1) { hv = h4[1]; rv = r4[1]; } else if (c == 2) { hv = h4[2]; rv = r4[2]; } else if (c == 3) { hv = h4[3]; rv = r4[3]; }
        rt[e * 2] = rol64(hv, (l - 1 - j) & 63);
        rt[e * 2 + 1] = rol64(rv, j);
    }
    __syncthreads();

    // ---- phase 1: ASCII -> 2-bit codes in LDS (coalesced 16-byte loads, 8 in flight per lane) ------
    {
        u32 bad_any = 0;
        constexpr int NCHUNK = (TILE + HALO) / 16;                 // 4104 = 16 * 256 + 8
        constexpr int H16 = HALO / 16;
        const bool interior = tile_start >= HALO && tile_start + TILE <= nb;
        if (interior) {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = (const u32x4*)(a.bases + (tile_start - HALO));      // chunk ci lives at src[ci]
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {                    // 4 x 16-byte loads in flight per lane, streamed once
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32x4 q = __builtin_nontemporal_load(src + tid + TILE_THREADS * (grp * 4 + u)); v[u] = make_uint4(q.x, q.y, q.z, q.w); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { u32 bad = 0; codes[tid + TILE_THREADS * (grp * 4 + u) - H16] = pack16(v[u], bad); bad_any |= bad; }
            }
            if (tid < NCHUNK - 16 * TILE_THREADS) { u32 bad = 0; const u32x4 q = src[16 * TILE_THREADS + tid]; codes[16 * TILE_THREADS + tid - H16] = pack16(make_uint4(q.x, q.y, q.z, q.w), bad); bad_any |= bad; }
        } else {
            // first / last tile of the batch: positions outside [0, n_bases) read as 'A'
#pragma unroll 1
            for (int ci = tid; ci < NCHUNK; ci += TILE_THREADS) {
                const int64_t pos = tile_start + 16 * (int64_t)(ci - H16);
                uint4 v = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
                if (pos >= 0 && pos + 16 <= nb) v = *(const uint4*)(a.bases + pos);
                else if (pos >= 0 && pos < nb) {
                    u32 w4[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
#pragma unroll
                    for (int i = 0; i < 16; ++i) if (pos + i < nb) w4[i >> 2] = (w4[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((u32)a.bases[pos + i] << (8 * (i & 3)));
                    v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
                u32 bad = 0;
                codes[ci - H16] = pack16(v, bad);
                bad_any |= bad;
            }
        }
        if (bad_any) {                       // some byte of my chunks is not one of ACGT
            misc[8] = 1;                     // whole tile takes the generic exact path
            for (int ci = tid; ci < NCHUNK; ci += TILE_THREADS) {
                const int64_t pos = tile_start + 16 * (int64_t)(ci - H16);
                for (int i = 0; i < 16; ++i) {
                    int64_t q = pos + i;
                    if (q >= 0 && q < nb) { u8 c = a.bases[q]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') *a.err_flag = 1; }
                }
            }
        }
    }
    __syncthreads();
    MDBG_STAMP(1);
    if (misc[8]) {
        if (tid == 0) { a.n_cand[t] = SLOW_MARK; a.slow_list[atomicAdd(a.slow_count, 1u)] = t; }
        return;
    }

    // ---- phase 2: per-lane rolling partial hashes over SEG raw bases ------------------------------
    const u32 thrF = a.c.thrF, thrR = a.c.thrR, bfe_off = a.c.bfe_off;
    RollConsts K;                             // asm outputs: they stay in VGPRs
    asm("v_mov_b32 %0, %1" : "=v"(K.hs_shift) : "s"(bfe_off - 5u));
    asm("v_mov_b32 %0, 0x60" : "=v"(K.k60));
    asm("v_mov_b32 %0, %1" : "=v"(K.maskR) : "s"(a.c.maskR));
    u32 cb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cb[i] = 0;
    bool force = false;                       // first l-1 pushes of my segment must all be candidates
    const bool active = tile_start + (int64_t)tid * SEG < nb;
    if (active) {
        RollState st;
        // warm-up: the same loop, silently, over the 32 bases before my segment (then the whole 128-base halo).  The
        // state starts as l x 'A' with matching G0/R0, so "state = hash of the last l pushed codes" holds from the first
        // push and is exact after l pushes.
        u32 npush = 0;
        for (int nW = 2;; nW = HALO / 16) {
            st.hist = 0; st.G = a.c.G0; st.R = a.c.R0; npush = 0;
            const int ws = tid * SEG_WORDS - nW;
            st.prev8 = (ws - 1 >= -HALO / 16) ? (codes[ws - 1] >> 30) << 3 : 32u;
            for (int k = 0; k < nW; ++k) {
                const u32 w = codes[ws + k];
                npush += __popc(roll8<HPC, false>(st, w << 3, tbl, K, thrF, thrR));
                npush += __popc(roll8<HPC, false>(st, w >> 13, tbl, K, thrF, thrR));
            }
            if (npush >= l || nW == HALO / 16) break;
        }
        if (npush < l) {
            // rare: fewer than l code changes in the 128 bases before my segment (long homopolymer).  Replay from global
            // memory, at most 4096 bases back; if that is still not enough, or a byte outside ACGT is met, the partial
            // hashes of my first l-1 pushes cannot be trusted -> force them to be candidates.
            const int64_t seg0 = tile_start + (int64_t)tid * SEG;
            int64_t q = seg0; u32 runs = 0; bool dirty = false;
            while (q > 0 && runs < l + 1 && seg0 - q < 4096) {
                --q;
                const u8 b0 = a.bases[q];
                if (b0 != 'A' && b0 != 'C' && b0 != 'G' && b0 != 'T') dirty = true;
                if (!HPC || q == 0 || (((u32)a.bases[q - 1] >> 1) & 3u) != (((u32)b0 >> 1) & 3u)) ++runs;
            }
            if (dirty || (runs < l + 1 && q > 0)) force = true;
            st.hist = 0; st.G = a.c.G0; st.R = a.c.R0;
            u32 prev = q > 0 ? (((u32)a.bases[q - 1] >> 1) & 3u) : 4u;
            for (; q < seg0; ++q) {
                const u32 c_ = ((u32)a.bases[q] >> 1) & 3u;
                if (!HPC || c_ != prev) {
                    const u32 c8_ = c_ << 3;
                    const u32 ad_ = (__builtin_amdgcn_ubfe(st.hist, bfe_off, 2u) << 5) | c8_;
                    st.hist = (st.hist << 2) | c8_;
                    const uint2 x_ = *(const uint2*)((const char*)tbl + ad_);
                    st.G = (st.G << 1) ^ x_.x; st.R = (st.R >> 1) ^ x_.y;
                }
                prev = c_;
            }
            st.prev8 = prev << 3;
        }
        // main loop: 8 iterations x 32 bases; cb[] is rotated so that every index stays static (registers)
        const uint2* segp = (const uint2*)(codes + tid * SEG_WORDS);
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
            const uint2 v = segp[it];
            u32 b0 = roll8<HPC, true>(st, v.x << 3, tbl, K, thrF, thrR);
            u32 b1 = roll8<HPC, true>(st, v.x >> 13, tbl, K, thrF, thrR);
            u32 b2 = roll8<HPC, true>(st, v.y << 3, tbl, K, thrF, thrR);
            u32 b3 = roll8<HPC, true>(st, v.y >> 13, tbl, K, thrF, thrR);
            // each bK holds 8 flags with step 0 in bit 7: concatenate (first step in the top bit) and reverse
            const u32 word = __brev((b0 << 24) | (b1 << 16) | (b2 << 8) | b3);
#pragma unroll
            for (int j = 0; j < 7; ++j) cb[j] = cb[j + 1];
            cb[7] = word;
        }
    }
    if (force) {                              // mark the first l-1 pushes of my segment
        u32 pushes = 0; u32 pv = codes[tid * SEG_WORDS - 1] >> 30;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            for (int b = 0; b < 32 && pushes < l - 1; ++b) {
                const int p = i * 32 + b;
                const u32 c = (codes[tid * SEG_WORDS + (p >> 4)] >> (2 * (p & 15))) & 3u;
                if (!HPC || c != pv) { cb[i] |= 1u << b; ++pushes; }
                pv = c;
            }
        }
    }

    MDBG_STAMP(2);
    // ---- phase 3: ordered candidate list -------------------------------------------------------------
    u32 cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) cnt += __popc(cb[i]);
    u32 n_cand;
    u32 base = block_excl_scan_256(cnt, misc, n_cand);
    if (n_cand > QCAP) {                      // too dense for the slab: generic path
        if (tid == 0) { a.n_cand[t] = SLOW_MARK; a.slow_list[atomicAdd(a.slow_count, 1u)] = t; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 m = cb[i];
        while (m) { const int b = __ffs(m) - 1; m &= m - 1; list[base++] = (u16)(tid * SEG + i * 32 + b); }
    }
    __syncthreads();

    MDBG_STAMP(3);
    // ---- phase 4: exact 64-bit fix-up of every candidate ----------------------------------------------
    const u32 rl = a.bread[gt], rhi = a.bread[gt + 1];
    const int64_t lds_lo = tile_start - HALO;            // first raw position staged in LDS
    auto code_at = [&](int64_t q) -> u32 { const int rel = (int)(q - tile_start); return (codes[rel >> 4] >> (2 * (rel & 15))) & 3u; };
    u32 nval = 0;
    Rec* slab = a.slab + (size_t)t * QCAP;
    for (u32 j = tid; j < n_cand; j += TILE_THREADS) {
        const int64_t p = tile_start + list[j];
        Rec rec; rec.hash = 0; rec.pos = 0; rec.read = 0xFFFFFFFFu;
        if (p < nb) {
            bool ok = true, in_lds = true;
            int64_t rlo = 0; u32 r = 0;
            // a run continuation (repeated verdicts, merged read starts) is never the end of a valid l-mer: a valid end is
            // a run start at HPC index >= l-1 >= 1 of its read, i.e. its code differs from the base before it
            if (HPC && code_at(p - 1) == code_at(p)) ok = false;
            u64 fh = 0, rh = 0; int64_t q = p;
            if (ok) {
                r = find_read(a.offsets, rl, rhi, (u64)p); rlo = (int64_t)a.offsets[r];
                if (p < rlo) ok = false;                     // lead-in bytes in front of the batch's first read belong to no read
                for (int jj = (int)l - 1;; --jj) {
                    const u32 c = code_at(q);
                    const ulonglong2 e = *(const ulonglong2*)(rt + (jj * 4 + c) * 2);
                    fh ^= e.x; rh ^= e.y;
                    if (jj == 0) break;
                    if (q == rlo) { ok = false; break; }
                    int64_t q2 = q - 1;
                    if (q2 < lds_lo) { in_lds = false; break; }
                    if (HPC) {
                        const u32 c2 = code_at(q2);
                        while (q2 > rlo) { if (q2 - 1 < lds_lo) { in_lds = false; break; } if (code_at(q2 - 1) != c2) break; --q2; }
                        if (!in_lds) break;
                    }
                    q = q2;
                }
            }
            u64 h = fh < rh ? fh : rh, start = (u64)q;
            if (ok && !in_lds) ok = walk_lmer_ascii<HPC>(a.bases, (u64)rlo, (u64)p, l, start, h);   // ran off the staged region
            if (ok && h <= a.c.bound) { rec.hash = h; rec.pos = (u32)(start - (u64)rlo); rec.read = r + a.read_base; ++nval; }
        }
        slab[j] = rec;
    }
    u32 tot;
    (void)block_excl_scan_256(nval, misc, tot);
    if (tid == 0) { a.n_cand[t] = n_cand; a.n_valid[t] = tot; }
    MDBG_STAMP(4);
#undef MDBG_STAMP
}

// ---- gather: slabs -> final position-ordered arrays ---------------------------------------------
__global__ __launch_bounds__(256) void gather_kernel(u32 n_tiles, const Rec* __restrict__ slab, const u32* __restrict__ n_cand,
                                                     const u64* __restrict__ tile_base, u64* __restrict__ out_hash,
                                                     u32* __restrict__ out_pos, u32* __restrict__ out_read, u64 out_cap) {
    const u32 t = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per tile
    if (t >= n_tiles) return;
    const u32 lane = threadIdx.x & 63;
    const u32 nc = n_cand[t];
    if (nc == SLOW_MARK || nc == 0) return;
    const Rec* s = slab + (size_t)t * QCAP;
    u64 base = tile_base[t];
    for (u32 j0 = 0; j0 < nc; j0 += 64) {
        const u32 j = j0 + lane;
        Rec r; r.read = 0xFFFFFFFFu;
        if (j < nc) r = s[j];
        const bool v = r.read != 0xFFFFFFFFu;
        const u64 m = __ballot(v);
        if (v) {
            const u64 idx = base + __popcll(m & ((1ull << lane) - 1));
            if (idx < out_cap) { out_hash[idx] = r.hash; out_pos[idx] = r.pos; out_read[idx] = r.read; }
        }
        base += __popcll(m);
    }
}

// exclusive scan of n_valid over the tiles of one launch (carry[0] in/out = running total), three small kernels:
// sums of 1024-tile blocks, scan of the block sums by one workgroup, per-block scan + base.
__global__ __launch_bounds__(1024) void tile_scan_sums_kernel(u32 n, const u32* __restrict__ n_valid, u64* __restrict__ block_sum) {
    __shared__ u32 ws[16];
    const u32 i = blockIdx.x * 1024 + threadIdx.x;
    u32 v = i < n ? n_valid[i] : 0;
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { u64 t = 0; for (int q = 0; q < 16; ++q) t += ws[q]; block_sum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void tile_scan_top_kernel(u32 n_blocks, u64* __restrict__ block_sum, u64* __restrict__ carry) {
    __shared__ u64 ws[16]; __shared__ u64 run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) run = carry[0];
    __syncthreads();
    for (u32 i0 = 0; i0 < n_blocks; i0 += 1024) {
        const u32 i = i0 + tid;
        const u64 v = i < n_blocks ? block_sum[i] : 0;
        u64 inc = v;
        for (int d = 1; d < 64; d <<= 1) { const u64 o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        u64 b = run, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) b += ws[q]; tot += ws[q]; }
        if (i < n_blocks) block_sum[i] = b + inc - v;
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) carry[0] = run;
}
__global__ __launch_bounds__(1024) void tile_scan_final_kernel(u32 n, const u32* __restrict__ n_valid, const u64* __restrict__ block_base, u64* __restrict__ tile_base) {
    __shared__ u32 ws[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 i = blockIdx.x * 1024 + tid;
    const u32 v = i < n ? n_valid[i] : 0;
    const u32 inc = wave_incl_scan(v);
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    u64 b = block_base[blockIdx.x];
    for (int q = 0; q < wv; ++q) b += ws[q];
    if (i < n) tile_base[i] = b + inc - v;
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
__global__ void acc_slow_kernel(const u32* __restrict__ slow_count, u64* __restrict__ slow_total) { *slow_total += *slow_count; }

// Error path only (a byte outside ACGTN was seen somewhere in the batch): the reference's exact rule — nthash panics iff a
// read whose HPC string has at least l bases holds such a byte (src/read.rs:157-174 decides the length).  One wave per
// read; *which = smallest offending read index (~0: none).
__global__ __launch_bounds__(256) void alphabet_rule_kernel(const u8* __restrict__ bases, const u64* __restrict__ off, u32 n_reads, u32 l, u32 hpc,
                                                            unsigned long long* __restrict__ which) {
    const u32 r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const u32 lane = threadIdx.x & 63;
    const u64 a = off[r], b = off[r + 1];
    u64 kept = 0; bool bad = false;
    for (u64 p = a + lane; p < b; p += 64) {
        const u8 c = bases[p];
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
        if (!hpc || p == a || !(c == bases[p - 1] && in_hpc_set(c))) ++kept;
    }
    for (int d = 32; d; d >>= 1) kept += __shfl_down(kept, d, 64);
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0 && any_bad && kept >= l) atomicMin(which, (unsigned long long)r);
}
void launch_alphabet_rule(const u8* bases, const u64* off, u32 n_reads, u32 l, bool hpc, unsigned long long* which, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(alphabet_rule_kernel, dim3((n_reads + 3) / 4), dim3(256), 0, s, bases, off, n_reads, l, hpc ? 1u : 0u, which);
}

// ---- host launchers -------------------------------------------------------------------------------
struct SketchLaunch {
    const u8* bases; u64 n_bases; const u64* offsets; u32 n_reads;
    u32* bread; u64 n_tiles_total;
    Rec* slab; u32* n_cand; u32* n_valid; u64* tile_base; u64* scan_tmp; u32* slow_list; u32* slow_count; u32* err_flag; u64* carry;
    u64* out_hash; u32* out_pos; u32* out_read; u64 out_cap;
    SketchConsts c; const u32* tbl; bool force_slow; u64* slow_total; u32 read_base; u64* dbg;
};

void launch_bread(const SketchLaunch& L, hipStream_t s) {
    const u64 n = L.n_tiles_total + 1;
    hipLaunchKernelGGL(bread_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, L.offsets, L.n_reads, L.n_tiles_total, L.bread);
}

// one chunk of tiles [tile0, tile0+n): tile kernel, slow count, scan, gather, slow write
void launch_sketch_chunk(const SketchLaunch& L, u64 tile0, u32 n, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end) {
    TileArgs a;
    a.bases = L.bases; a.n_bases = L.n_bases; a.offsets = L.offsets; a.n_reads = L.n_reads; a.bread = L.bread;
    a.tile0 = tile0; a.n_tiles = n; a.slab = L.slab; a.n_cand = L.n_cand; a.n_valid = L.n_valid;
    a.slow_list = L.slow_list; a.slow_count = L.slow_count; a.err_flag = L.err_flag; a.c = L.c; a.tbl = L.tbl; a.read_base = L.read_base; a.dbg = L.dbg;
    (void)hipMemsetAsync(L.slow_count, 0, sizeof(u32), s);
    const bool hpc = L.c.hpc != 0;
    if (L.force_slow || L.c.l > (u32)FAST_MAX_L) {
        hipLaunchKernelGGL(all_slow_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, L.n_cand, L.slow_list, L.slow_count);
    } else {
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        if (hpc) hipLaunchKernelGGL(sketch_tile_kernel<true>, dim3(n), dim3(TILE_THREADS), 0, s, a);
        else     hipLaunchKernelGGL(sketch_tile_kernel<false>, dim3(n), dim3(TILE_THREADS), 0, s, a);
        if (ev_end) (void)hipEventRecord(ev_end, s);
    }
    if (hpc) hipLaunchKernelGGL((slow_tile_kernel<true, false>), dim3(n < 4096 ? n : 4096), dim3(256), 0, s, a, L.tile_base, L.out_hash, L.out_pos, L.out_read, L.out_cap);
    else     hipLaunchKernelGGL((slow_tile_kernel<false, false>), dim3(n < 4096 ? n : 4096), dim3(256), 0, s, a, L.tile_base, L.out_hash, L.out_pos, L.out_read, L.out_cap);
    const u32 nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(tile_scan_sums_kernel, dim3(nb), dim3(1024), 0, s, n, L.n_valid, L.scan_tmp);
    hipLaunchKernelGGL(tile_scan_top_kernel, dim3(1), dim3(1024), 0, s, nb, L.scan_tmp, L.carry);
    hipLaunchKernelGGL(tile_scan_final_kernel, dim3(nb), dim3(1024), 0, s, n, L.n_valid, L.scan_tmp, L.tile_base);
    hipLaunchKernelGGL(gather_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, L.slab, L.n_cand, L.tile_base, L.out_hash, L.out_pos, L.out_read, L.out_cap);
    if (hpc) hipLaunchKernelGGL((slow_tile_kernel<true, true>), dim3(n < 4096 ? n : 4096), dim3(256), 0, s, a, L.tile_base, L.out_hash, L.out_pos, L.out_read, L.out_cap);
    else     hipLaunchKernelGGL((slow_tile_kernel<false, true>), dim3(n < 4096 ? n : 4096), dim3(256), 0, s, a, L.tile_base, L.out_hash, L.out_pos, L.out_read, L.out_cap);
    hipLaunchKernelGGL(acc_slow_kernel, dim3(1), dim3(1), 0, s, L.slow_count, L.slow_total);
}

void launch_read_offsets(const u32* mread, u64 m0, u64 m1, u32 slot0, u32 n_reads, u64* off, hipStream_t s) {
    const u64 n = m1 - m0 + 1;
    hipLaunchKernelGGL(read_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mread, m0, m1, slot0, n_reads, off);
}

// Host side: constants of the candidate filter.  Codes: A=0 C=1 T=2 G=3.
void make_sketch_consts(u32 l, double density, bool hpc, SketchConsts& c) {
    double v = density * 18446744073709551616.0;              // src/read.rs:183
    u64 bound = !(v > 0.0) ? 0 : (v >= 18446744073709551616.0 ? ~0ull : (u64)v);
    c.bound = bound; c.l = l; c.hpc = hpc ? 1 : 0;
    const u64 h[4] = {NT_SEED_A, NT_SEED_C, NT_SEED_T, NT_SEED_G};
    const u64 rc[4] = {NT_SEED_T, NT_SEED_G, NT_SEED_A, NT_SEED_C};
    const u32 ll = l <= (u32)FAST_MAX_L ? l : (u32)FAST_MAX_L;    // constants only used by the fast kernel
    const u32 bh = (u32)(bound >> 32);
    c.thrF = bh | ((1u << (ll - 1)) - 1u);
    c.thrR = bh >> (ll - 1);
    c.maskR = (u32)((1ull << (33 - ll)) - 1ull);
    c.bfe_off = 2 * ll + 1;
    u32 G0 = 0, R0 = 0;
    for (u32 t = 0; t < ll; ++t) { G0 ^= (u32)(h[0] >> 32) << (ll - 1 - t); R0 ^= (u32)(rc[0] >> 32) >> (ll - 1 - t); }
    c.G0 = G0; c.R0 = R0;
    for (u32 o = 0; o < 4; ++o) for (u32 i = 0; i < 4; ++i) {
        const u32 uFo = (u32)(h[o] >> 32), uFi = (u32)(h[i] >> 32), uRo = (u32)(rc[o] >> 32), uRi = (u32)(rc[i] >> 32);
        c.tbl[2 * (o * 4 + i)] = (ll < 32 ? (uFo << ll) : 0u) ^ uFi;
        c.tbl[2 * (o * 4 + i) + 1] = (uRo >> ll) ^ uRi;
    }
}
