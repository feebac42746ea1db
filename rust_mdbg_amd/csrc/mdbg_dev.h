// mdbg_dev.h — shared types, constants and device helpers for libmdbg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---- tile geometry of the sketch kernel (sketch.hip) ----------------------------------------------
// A tile is staged by NW waves of 64 lanes, four words of 32 raw bases per lane: HALO bases of look-back (owned by the tile in
// front) followed by the STRIDE bases whose l-mer END positions the tile owns.  Two geometries are compiled:
//   NW = 1  one WAVE per tile (8,192 bases staged, 128 of look-back): no workgroup barrier anywhere in the tile's life, the waves of
//           a CU drift apart and their phases (compaction: VALU, filter: VALU + crossbar, exact evaluation: LDS + scalar) overlap;
//   NW = 4  one 256-lane workgroup per tile (32,768 staged, 256 of look-back): rounds 1-3; still used by the syncmer scheme.
// The host picks one per context (TileShape); everything behind the tile kernel (slabs, scans, gather) only sees tile counts.
constexpr int TILE_WPT = 4;                       // raw words per lane
template <int NW> struct TileGeo {
    static constexpr int THREADS = 64 * NW;
    static constexpr int RAW_WORDS = TILE_WPT * THREADS;
    static constexpr int HALO_BASES = NW == 1 ? 128 : 256;
    static constexpr int STRIDE = RAW_WORDS * 32 - HALO_BASES;       // NW = 4: 32,512 raw bases per tile; NW = 1: 8,064
};
struct TileShape { u32 nw, tpw, stride, halo; };   // tpw: tiles per workgroup of the launch (NW = 1: 1 or 4 waves per workgroup)
constexpr int MDBG_MAX_L_DEV = 32;                // = MDBG_MAX_L of the C ABI

// the table slots keep the A smallest ordinals of a k-min-mer for A up to this; larger min_abundance values get the A-th sighting from a
// re-scan of the windows at finalize (table.hip, wrap_list_kernel)
constexpr u32 MDBG_CASCADE_MAX = 8;
__host__ __device__ inline u32 cascade_of(u32 A) { return A <= MDBG_CASCADE_MAX ? A : 1u; }

// ordinal = (global read ordinal << WIN_BITS) | window index within the read
constexpr int WIN_BITS = 26;
constexpr u64 WIN_MASK = (1ull << WIN_BITS) - 1;

struct __attribute__((aligned(16))) Rec {         // one slab slot / one selected minimizer
    u64 hash;
    u32 pos;                                      // raw position relative to the read start
    u32 read;                                     // batch-local read index, 0xFFFFFFFF = rejected candidate
};

// ntHash seeds by 2-bit code ((ascii >> 1) & 3: A=0 C=1 T=2 G=3); nthash crate 0.5.1 H_LOOKUP / RC_LOOKUP
#define NT_SEED_A 0x3c8bfbb395c60474ull
#define NT_SEED_C 0x3193c18562a02b4cull
#define NT_SEED_G 0x20323ed082572324ull
#define NT_SEED_T 0x295549f54be24456ull

__host__ __device__ inline u64 rol64(u64 x, unsigned r) { r &= 63; return (x << r) | (x >> ((64 - r) & 63)); }

// ntHash contribution of an ASCII byte; N -> 0; any other byte -> 0 (the error is raised elsewhere)
__device__ inline u64 nt_h_ascii(u8 c) {
    return c == 'A' ? NT_SEED_A : c == 'C' ? NT_SEED_C : c == 'G' ? NT_SEED_G : c == 'T' ? NT_SEED_T : 0ull;
}
__device__ inline u64 nt_rc_ascii(u8 c) {
    return c == 'A' ? NT_SEED_T : c == 'C' ? NT_SEED_G : c == 'G' ? NT_SEED_C : c == 'T' ? NT_SEED_A : 0ull;
}
// src/read.rs:163 — only these bytes collapse into homopolymer runs
__device__ inline bool in_hpc_set(u8 c) {
    return c == 'A' || c == 'C' || c == 'T' || c == 'G' || c == 'a' || c == 'c' || c == 't' || c == 'g' || c == 'N' || c == 'n';
}

// ---- wave / block scans (wave = 64 lanes) -------------------------------------------------------
// inclusive scan over the 64 lanes with DPP row shifts / broadcasts (no LDS round trips): 4 steps inside each row of 16,
// then row 15 -> rows 1/3 and lane 31 -> rows 2,3
__device__ inline u32 wave_incl_scan(u32 v) {
#define MDBG_DPP_ADD(ctrl, rmask) v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, true)
    MDBG_DPP_ADD(0x111, 0xF);      // row_shr:1
    MDBG_DPP_ADD(0x112, 0xF);      // row_shr:2
    MDBG_DPP_ADD(0x114, 0xF);      // row_shr:4
    MDBG_DPP_ADD(0x118, 0xF);      // row_shr:8
    MDBG_DPP_ADD(0x142, 0xA);      // row_bcast:15 into rows 1 and 3
    MDBG_DPP_ADD(0x143, 0xC);      // row_bcast:31 into rows 2 and 3
#undef MDBG_DPP_ADD
    return v;
}
// exclusive scan over a 256-thread block; tmp must hold 5 u32 in LDS; every thread gets `total`
__device__ inline u32 block_excl_scan_256(u32 v, u32* tmp, u32& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 inc = wave_incl_scan(v);
    __syncthreads();                       // tmp may still be read by a previous call
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { u32 t = tmp[i]; if (i < w) base += t; tot += t; }
    total = tot;
    return base + inc - v;
}

// ---- the same two primitives for a tile of NW waves --------------------------------------------------
// NW = 1: the tile is one wave.  LDS instructions of one wave execute in issue order, so a store is visible to every later load of the
// same wave whichever lane issued it: "barrier" = keep the compiler from moving LDS accesses across it, no instruction at all.
template <int NW> __device__ __forceinline__ void tile_sync() {
    if constexpr (NW == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
template <int NW> __device__ __forceinline__ u32 tile_excl_scan(u32 v, u32* tmp, u32& total) {
    if constexpr (NW == 1) {
        const u32 inc = wave_incl_scan(v);
        total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        return inc - v;
    } else {
        static_assert(NW == 4, "block scan over 256 threads");
        return block_excl_scan_256(v, tmp, total);
    }
}

// Sharded counters.  Same-address device atomics serialise at ~12 ns each on MI355X (MI355X_MICROARCH.md "fanin"), so
// a hot counter is spread over CTR_SHARDS addresses chosen by (block, wave); sum_shards_kernel folds them when the
// host needs the value.
constexpr int CTR_SHARDS = 4096;
__device__ inline u64* ctr_shard(u64* ctr) { return ctr + (((u32)blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u >> 20); }
// every ACTIVE lane calls this; adds the number of active lanes with one atomic per wave
__device__ inline void wave_agg_inc(u64* ctr) {
    const u64 act = __ballot(1);
    const int lane = threadIdx.x & 63;
    if (lane == __ffsll((unsigned long long)act) - 1) atomicAdd((unsigned long long*)ctr_shard(ctr), (unsigned long long)__popcll(act));
}
// adds popcount(pred over the wave) with one atomic; must be called by all lanes of the wave (convergent)
__device__ inline void wave_count_add(bool pred, u64* ctr) {
    const u64 m = __ballot(pred);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long*)ctr_shard(ctr), (unsigned long long)__popcll(m));
}

// 64-bit finaliser (murmur3 fmix64)
__host__ __device__ inline u64 fmix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
